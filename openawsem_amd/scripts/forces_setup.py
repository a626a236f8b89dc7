"""Default force list, same content and order as the reference's user-editable openawsem/scripts/forces_setup.py
(:33-77).  A project's own forces_setup.py written for the reference (``from openawsem.functionTerms import *``)
is loaded unmodified by awsem_run / awsem_analyze after `openawsem_amd.install_as_openawsem()`."""
from openawsem_amd.functionTerms import *  # noqa: F401,F403
from openawsem_amd.functionTerms import (basicTerms, biasTerms, contactTerms, debyeHuckelTerms, hydrogenBondTerms,
                                         templateTerms)
from openawsem_amd.unit import angstrom


def set_up_forces(oa, computeQ=False, submode=-1, contactParameterLocation=".", membrane_center=-0 * angstrom):
    forces = [
        basicTerms.con_term(oa),
        basicTerms.chain_term(oa),
        basicTerms.chi_term(oa),
        basicTerms.excl_term(oa, periodic=False),
        basicTerms.rama_term(oa),
        basicTerms.rama_proline_term(oa),
        contactTerms.contact_term(oa),
        hydrogenBondTerms.beta_term_1(oa),
        hydrogenBondTerms.beta_term_2(oa),
        hydrogenBondTerms.beta_term_3(oa),
        hydrogenBondTerms.pap_term_1(oa),
        hydrogenBondTerms.pap_term_2(oa),
        templateTerms.fragment_memory_term(oa, frag_file_list_file="./single_frags.mem", npy_frag_table="./single_frags.npy",
                                           UseSavedFragTable=False),
        debyeHuckelTerms.debye_huckel_term(oa, chargeFile="charge.txt"),
    ]
    if computeQ:
        forces.append(biasTerms.rg_term(oa))
        forces.append(biasTerms.q_value(oa, "crystal_structure-cleaned.pdb", forceGroup=1))
    return forces
