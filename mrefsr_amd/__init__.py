"""mrefsr_amd -- MI355X (gfx950) implementation of MRefSR's multi-reference
matching-and-reconstruction hot path behind the reference's own interfaces.

    mrefsr_amd.csrc/            hand-written HIP kernels + the C ABI (include/mrefsr_hip.h)
    mrefsr_amd.hip              torch-tensor level bindings of the C ABI (ctypes, no torch in the .so)
    mrefsr_amd.ops.{dcn,fused_act,upfirdn2d}   mirror of basicsr.ops.* (same names / arguments)
    mrefsr_amd.archs.*          mirror of the basicsr.archs modules on the path, ARCH_REGISTRY
    mrefsr_amd.models.*         MultiRefRestorationModel (feed_data / optimize_parameters / test)

There is no CPU fallback: every op raises if the HIP library is missing or a tensor is not on the
GPU (the reference's own native ops do the same: basicsr/ops/dcn/deform_conv.py:61-62,143-144).
"""
__version__ = '0.1.0'
