"""Live re-check of the oracle against the reference itself (dev container only: skipped when
/root/reference or oracle/_ref is absent, e.g. on the GPU box)."""
import numpy as np
import pytest

import ref_loader
from waterorderlib_amd import synth

pytestmark = pytest.mark.skipif(not ref_loader.reference_available(),
                                reason="reference not present (run oracle/build_ref.sh in the dev container)")


@pytest.mark.parametrize("n,seed", [(300, 11), (1200, 12)])
def test_q_matches_reference(oracle, n, seed):
    wp, wl = ref_loader.load_reference()
    xyz, boxes = synth.water_frames(n, 1, seed=seed)
    pos, box = xyz[0], boxes[0]
    q_ref = wp.getOrderParamq(pos, pos, box)
    q = oracle.getOrderParamq(pos, pos, box)
    np.testing.assert_allclose(q, q_ref, rtol=0, atol=1e-12)
    assert np.array_equal(oracle.allnearneighbors(pos, box, 0.0, 10.0), wl.allnearneighbors(pos, box, 0.0, 10.0))
    ca_ref, cn_ref = wp.getCosAngs(pos, pos, box)
    ca, cn = oracle.getCosAngs(pos, pos, box)
    assert np.array_equal(cn, cn_ref)
    np.testing.assert_allclose(ca, ca_ref, rtol=0, atol=1e-10)
    li_ref, ln_ref = wp.getLSI(pos, pos, box)
    li, ln = oracle.getLSI(pos, pos, box)
    assert np.array_equal(ln, ln_ref)
    np.testing.assert_allclose(li, li_ref, rtol=0, atol=1e-12)
