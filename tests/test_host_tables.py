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


def test_backbone_variants_without_cb_springs():
    """chain_no_cb_constraint_term / con_no_cb_constraint_term (reference basicTerms.py:542-615) are subsets of the chain /
    con springs: N-C of interior residues only; CA-CB only at chain ends."""
    from openawsem_amd.functionTerms import basicTerms
    from tests.common import golden_system
    oa, _, _ = golden_system("1brs")          # three chains
    def bonds(f):
        return {p: (r0, k) for p, r0, k in zip(f.pairs, f.r0, f.k)}
    chain, chain_no = bonds(basicTerms.chain_term(oa)), bonds(basicTerms.chain_no_cb_constraint_term(oa))
    con, con_no = bonds(basicTerms.con_term(oa)), bonds(basicTerms.con_no_cb_constraint_term(oa))
    interior = [i for i in range(oa.nres) if i not in oa.chain_starts and i not in oa.chain_ends]
    assert set(chain_no) == {(oa.n[i], oa.c[i]) for i in interior}
    assert all(chain[b] == chain_no[b] for b in chain_no)
    ends = [i for i in range(oa.nres) if (i in oa.chain_starts or i in oa.chain_ends) and oa.res_type[i] != "IGL"]
    cb_bonds = {(oa.ca[i], oa.cb[i]) for i in range(oa.nres) if oa.res_type[i] != "IGL"}
    assert set(con_no) & cb_bonds == {(oa.ca[i], oa.cb[i]) for i in ends}
    assert set(con) - cb_bonds == set(con_no) - cb_bonds and all(con[b] == con_no[b] for b in con_no)


def test_excl_term_v2_pair_rule(oracle_api):
    """excl_term_v2 (reference basicTerms.py:194-263): the pair set follows its step() expression, evaluated here for
    every pair of the interaction groups, and the oracle's energy equals the direct sum over that set."""
    from openawsem_amd.forces import System
    from openawsem_amd.functionTerms import basicTerms
    from oracle import awsem_oracle
    from tests.common import golden_system
    oa, traj, _ = golden_system("1brs")
    f = basicTerms.excl_term_v2(oa)
    ca, cb, o = list(oa.ca), [x for x in oa.cb if x > 0], list(oa.o)
    edge, ca_set = set(oa.chain_ends) | set(oa.chain_starts), set(oa.ca)
    x = traj[9]
    groups = [(ca, ca), (cb, cb), (ca, cb), (o, o)]
    seen, want, left_out = set(), 0.0, set()
    for g1, g2 in groups:
        for a in g1:
            for b in g2:
                key = (min(a, b), max(a, b))
                if a == b or key in seen:
                    continue
                seen.add(key)
                s = abs(oa.resi[a] - oa.resi[b]) - 2 + (a in edge) * (b in edge) + (a not in ca_set) + (b not in ca_set)
                if s < 0:
                    left_out.add(key)
                    continue
                r = np.linalg.norm(x[a] - x[b])
                if r < 0.35:
                    want += 8368 * (r - 0.35) ** 2
    assert set(f.excl_pairs) == left_out and len(left_out) > oa.nres
    system = System(oa.system.masses.copy())
    system.addForce(f)
    cpu = awsem_oracle.make_context(system, n_replicas=1)
    cpu.set_positions(x, replica=0)
    assert abs(cpu.energy_by_group()[0, 20] - want) < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("cell", ["0", "1"])
def test_excl_term_v2_on_the_gpu(cell, monkeypatch):
    from openawsem_amd import _capi
    from openawsem_amd.forces import System
    from openawsem_amd.functionTerms import basicTerms
    from oracle import awsem_oracle
    from tests.common import golden_system
    monkeypatch.setenv("AWSEM_FORCE_CELL", cell)
    oa, traj, _ = golden_system("1brs")
    system = System(oa.system.masses.copy())
    system.virtual_sites = dict(oa.system.virtual_sites)
    system.addForce(basicTerms.excl_term_v2(oa))
    gpu = _capi.Context(_capi.load_hip_library(), oa.natoms, n_replicas=2, precision=1)
    system.upload(gpu)
    cpu = awsem_oracle.make_context(system, n_replicas=2)
    for r, k in enumerate((2, 9)):
        gpu.set_positions(traj[k], replica=r)
        cpu.set_positions(traj[k], replica=r)
    eg, ec = gpu.energy_by_group()[:, 20], cpu.energy_by_group()[:, 20]
    assert np.abs(ec).min() > 0.01 and np.abs(eg - ec).max() <= 1e-9 * np.abs(ec).max() + 1e-8
    assert np.abs(gpu.forces(redistribute=False) - cpu.forces(redistribute=False)).max() <= 1e-6
    gpu.close()
