"""CPU oracle for the MRefSR hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package.  Nothing under ``mrefsr_amd/`` does (tests/test_boundary.py greps for it).

``oracle.c_api``   ctypes binding of oracle/mrefsr_oracle.c (numpy in / numpy out)
``oracle.dcn_torch``  pure-torch DCNv2 (autograd-capable) restating the vendored CUDA spec
``oracle.pipeline``   torch-CPU restatement of extractor -> correspondence -> restoration net
"""
from . import c_api  # noqa: F401
