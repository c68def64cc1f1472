"""Mirror of the reference's ``basicsr.ops`` package (dcn, fused_act, upfirdn2d): same public
names, argument order and error behaviour; the native side is libmrefsr_hip.so."""
