#!/usr/bin/env python3
"""Register / LDS / scratch use of every kernel in a device assembly file (hipcc -S --cuda-device-only):
   python tools/kernel_regs.py file.s [name filter]"""
import re
import subprocess
import sys
s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ''
for b in s.split('  - .agpr_count:')[1:]:
    name = re.search(r'\.name:\s+(\S+)', b).group(1)
    dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    if flt not in dem:
        continue
    g = lambda k: re.search(r'\.%s:\s+(\d+)' % k, b).group(1)
    short = dem.replace('void ', '').replace('(anonymous namespace)::', '')
    short = short[:short.find('>(') + 1] if '>(' in short else short.split('(')[0]
    print(f"agpr {b.split(chr(10))[0].strip():>3} vgpr {g('vgpr_count'):>3} sgpr {g('sgpr_count'):>3} lds {g('group_segment_fixed_size'):>6} "
          f"scratch {g('private_segment_fixed_size'):>4}  {short[:90]}")
