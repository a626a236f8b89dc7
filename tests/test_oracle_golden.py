"""Pins the CPU oracle (and the host-side term factories feeding it) to the reference's golden energies:
tests/data/<protein>_energies.csv of the reference = OpenMM Reference-platform (fp64) energies of 12 trajectory
frames per protein, 7 non-zero columns, written with %.6f; the reference's own tolerance is atol=1e-5
(tests/test_energies.py:189,196)."""
import numpy as np
import pytest

from oracle import awsem_oracle
from tests.common import COLUMNS, PROTEINS, energies_by_column, golden_system


@pytest.mark.parametrize("protein", PROTEINS)
def test_oracle_matches_reference_golden_energies(protein, oracle_api):
    oa, traj, golden = golden_system(protein)
    assert traj.shape == (12, oa.natoms, 3)
    ctx = awsem_oracle.make_context(oa.system, n_replicas=len(traj), threads=4)
    got = energies_by_column(ctx, traj)
    for column in COLUMNS:
        assert np.allclose(got[column], golden[column], atol=1e-5, rtol=0), \
            f"{protein} {column}: max |diff| = {np.abs(got[column] - golden[column]).max():.3e}"
    # the columns the reference lists but never populates stay identically zero
    for column in ("Membrane", "ER", "TBM_Q"):
        assert np.all(got[column] == 0.0)
