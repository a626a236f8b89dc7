"""The vectorised builders of the hydrogen-bond tables against the reference's own double loops
(hydrogenBondTerms.py:216-359, 513-519, 577-583, 650-655, 722-727): entries must be bit-identical."""
import numpy as np
import pytest

from openawsem_amd.functionTerms import hydrogenBondTerms as hb
from openawsem_amd.functionTerms.contactTerms import _chain_index
from tests.common import golden_system


@pytest.mark.parametrize("protein", ["1brs", "1ubq"])      # 1brs has two chains
def test_lambda_and_gamma_tables_equal_the_scalar_loops(protein):
    oa, _, _ = golden_system(protein)
    for fn in (hb.get_Lambda_2, hb.get_Lambda_3):
        fast, slow = hb._lambda_table(oa, fn, block=37), hb._lambda_table_scalar(oa, fn)
        assert np.count_nonzero(slow) > 100 and np.array_equal(fast, slow), fn.__name__
    nres = oa.nres
    rng = np.random.default_rng(0)
    ssweight = np.zeros((nres, 2))
    ssweight[rng.random(nres) < 0.4, 1] = 1.0
    chain = [_chain_index(i, oa.chain_ends) for i in range(nres)]
    g1 = np.array([[hb.get_pap_gamma_APH(i, j, chain[i], chain[j], 1.0) + hb.get_pap_gamma_AP(i, j, chain[i], chain[j], 0.4, ssweight)
                    for j in range(nres)] for i in range(nres)])
    g2 = np.array([[hb.get_pap_gamma_P(i, j, chain[i], chain[j], 0.4, ssweight) for j in range(nres)] for i in range(nres)])
    assert np.array_equal(hb._pap_gamma_table(oa, ssweight, parallel=False, block=41), g1)
    assert np.array_equal(hb._pap_gamma_table(oa, ssweight, parallel=True, block=41), g2)
