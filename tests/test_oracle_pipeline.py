"""CPU: the torch-CPU restatement of the whole forward path (oracle/pipeline.py) reproduces the
output of the reference's own MultiRefRestorationModel.test() (golden e2e)."""
import numpy as np
import torch

import synth
from conftest import spec_from
from oracle import pipeline


def test_pipeline_oracle_matches_reference_model(golden):
    g = golden('e2e')
    sds = {n: synth.state_dict(spec_from(g, n + '_')) for n in ('net_g', 'net_extractor', 'net_map')}
    data = {k: torch.from_numpy(g[k]) for k in ('img_in_lq', 'img_in_up', 'img_ref_list')}
    torch.set_num_threads(8)
    outs = {}
    for fast in (True, False):  # C im2col + GEMM, and the pure-torch gather formulation of the DCN
        pipeline.FAST_DCN = fast
        out, idx = pipeline.forward(sds['net_g'], sds['net_extractor'], sds['net_map'], data)
        np.testing.assert_array_equal(idx, g['max_idx'])
        np.testing.assert_allclose(out.numpy(), g['out_test'], rtol=0, atol=2e-5)
        outs[fast] = out
    pipeline.FAST_DCN = True
    # the restatement's net_g on GIVEN matches (what the GPU parity test feeds it when a near-tie flips): same bits as matching itself
    out2, idx2 = pipeline.forward(sds['net_g'], sds['net_extractor'], sds['net_map'], data, max_idx=g['max_idx'])
    np.testing.assert_array_equal(idx2, g['max_idx'])
    assert torch.equal(out2, outs[True])
