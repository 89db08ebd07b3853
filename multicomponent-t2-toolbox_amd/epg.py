"""Drop-ins for epg/epg.py: the EPG dictionary is built on the device (one wavefront per
(T2, flip angle)) and returned in the reference's layout."""
import numpy as np

from .plan import Met2Plan


def create_Dic_3D(Npc, T2s, T1s, nEchoes, tau, alpha_values, TR):
    """epg/epg.py:155-162 -> Dic_3D[nEchoes, Npc, len(alpha_values)]"""
    alpha_values = np.atleast_1d(np.asarray(alpha_values, dtype=np.float64))
    plan = Met2Plan(int(nEchoes), int(Npc), alpha_values.shape[0])
    try:
        plan.build_dictionary_epg(T2s, T1s, tau, alpha_values, TR)
        return plan.get_dictionary()
    finally:
        plan.close()


def create_met2_design_matrix_epg(Npc, T2s, T1s, nEchoes, tau, flip_angle, TR):
    """epg/epg.py:47-62 -> design_matrix[nEchoes, Npc]"""
    return create_Dic_3D(Npc, T2s, T1s, nEchoes, tau, [flip_angle], TR)[:, :, 0]
