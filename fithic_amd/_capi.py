"""ctypes binding of libfithic_mi355x.so (the C ABI declared in include/fithic_mi355x.h).

The product path has NO CPU fallback: if the shared library is missing this module raises at import of
the symbols, and every kernel entry point fails loudly (FHX_ERR_NO_DEVICE) without a GPU.
"""
import ctypes
import os
import subprocess

import numpy as np

from . import _loader

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = _loader.LIB_PATH
CSRC = os.path.join(_PKG, "csrc")

FHX_OK = 0
FHX_ERR_ARG, FHX_ERR_NO_DEVICE, FHX_ERR_HIP, FHX_ERR_UNSUPPORTED, FHX_ERR_REFERENCE_EXIT, FHX_ERR_NOMEM, FHX_ERR_INTERNAL = -1, -2, -3, -4, -5, -6, -7
MODE_INTRA_ONLY, MODE_INTER_ONLY, MODE_ALL = 0, 1, 2
# FHX_TOTALS_*: how bdtrc sees a total of counts that does not fit a C int (include/fithic_mi355x.h; fithic.py:1070, 1101)
TOTALS_REFERENCE, TOTALS_WIDE = 0, 1
INT64_MAX = (1 << 63) - 1

# enum fhx_array
(A_HIST_SUMCC, A_HIST_NPAIRS, A_BIN_LB, A_BIN_UB, A_BIN_POSS, A_BIN_SUMCC, A_BIN_SUMDIST, A_BIN_POSS7, A_X, A_Y,
 A_KNOTS, A_COEFFS, A_TABLE_X, A_TABLE_Y0, A_TABLE_Y, A_OUTLIER_DIST_HIST, A_FDR_COUNTS, A_BIN_POSS0, A_DIST_KEYS,
 A_OUTLIER_DISTS) = range(20)
_ARRAY_DTYPE = {A_BIN_SUMDIST: np.float64, A_X: np.float64, A_Y: np.float64, A_KNOTS: np.float64,
                A_COEFFS: np.float64, A_TABLE_Y0: np.float64, A_TABLE_Y: np.float64}


class FhxParams(ctypes.Structure):
    _fields_ = [("resolution", ctypes.c_int64), ("dist_low", ctypes.c_int64), ("dist_up", ctypes.c_int64),
                ("n_bins", ctypes.c_int32), ("mapp_thres", ctypes.c_int32), ("mode", ctypes.c_int32),
                ("totals", ctypes.c_int32), ("bias_low", ctypes.c_double), ("bias_up", ctypes.c_double)]


class FhxStats(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int64) for n in ("n_rows", "inter_count", "inter_sum", "intra_all_count", "intra_all_sum",
                                              "in_range_count", "in_range_sum", "max_count", "n_dist", "n_skipped")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class FhxFitInfo(ctypes.Structure):
    _fields_ = [("n_bins_made", ctypes.c_int32), ("n_knots", ctypes.c_int32), ("spline_ier", ctypes.c_int32),
                ("spline_restarted", ctypes.c_int32), ("n_table", ctypes.c_int64), ("n_frags", ctypes.c_int64),
                ("possible_intra_in_range", ctypes.c_int64), ("possible_inter_all", ctypes.c_double),
                ("possible_intra_all", ctypes.c_double), ("max_possible_dist", ctypes.c_double),
                ("inter_chr_prob", ctypes.c_double), ("baseline_intra_prob", ctypes.c_double),
                ("spline_s", ctypes.c_double), ("spline_fp", ctypes.c_double), ("residual", ctypes.c_double),
                ("bh_total_tests", ctypes.c_double), ("outlier_thres", ctypes.c_double),
                ("totals", ctypes.c_int32), ("totals_narrowed", ctypes.c_int32), ("bdtrc_n_intra", ctypes.c_int64),
                ("bdtrc_n_inter", ctypes.c_int64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


# every symbol include/fithic_mi355x.h declares: name -> (restype, argtypes)
_P = ctypes.c_void_p
_I32P = ctypes.POINTER(ctypes.c_int32)
_I64P = ctypes.POINTER(ctypes.c_int64)
_F64P = ctypes.POINTER(ctypes.c_double)
SYMBOLS = {
    "fhx_create": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(_P)]),
    "fhx_destroy": (None, [_P]),
    "fhx_last_error": (ctypes.c_char_p, [_P]),
    "fhx_version": (ctypes.c_char_p, []),
    "fhx_warmup": (ctypes.c_int, [ctypes.c_int]),
    "fhx_set_params": (ctypes.c_int, [_P, ctypes.POINTER(FhxParams)]),
    "fhx_run_pass": (ctypes.c_int, [_P, ctypes.POINTER(FhxStats), ctypes.POINTER(FhxFitInfo)]),
    "fhx_load_fragments": (ctypes.c_int, [_P, _I32P, _I32P, _I32P, ctypes.c_int64, _I32P, ctypes.c_int32]),
    "fhx_host_inflate": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int32, ctypes.POINTER(_P)]),
    "fhx_text_bytes": (ctypes.c_int64, [_P]),
    "fhx_text_copy": (ctypes.c_int, [_P, _P, ctypes.c_int64]),
    "fhx_text_error": (ctypes.c_char_p, [_P]),
    "fhx_host_parse_text": (ctypes.c_int, [_P, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(_P)]),
    "fhx_text_free": (None, [_P]),
    "fhx_ingest_contacts_text": (ctypes.c_int, [_P, _P, ctypes.c_int32, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int32)]),
    "fhx_ingest_contacts_name": (ctypes.c_char_p, [_P, ctypes.c_int32]),
    "fhx_ingest_contacts_file": (ctypes.c_int, [_P, ctypes.c_char_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int32),
                                                ctypes.POINTER(ctypes.c_int32)]),
    "fhx_debug_inflate_file": (ctypes.c_int, [_P, ctypes.c_char_p, _P, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64)]),
    "fhx_ingest_contacts_commit": (ctypes.c_int, [_P, _I32P, ctypes.c_int32]),
    "fhx_ingest_contacts_discard": (None, [_P]),
    "fhx_fetch_pairs": (ctypes.c_int, [_P, ctypes.POINTER(ctypes.c_int64), ctypes.c_int64, _I32P, _I32P, _I32P, _I32P, _I32P]),
    "fhx_load_bias": (ctypes.c_int, [_P, _I32P, _I32P, _F64P, ctypes.c_int64]),
    "fhx_load_pairs": (ctypes.c_int, [_P, _I32P, _I32P, _I32P, _I32P, _I32P, ctypes.c_int64]),
    "fhx_load_pairs_device": (ctypes.c_int, [_P, _P, _P, _P, _P, _P, ctypes.c_int64, _P]),
    "fhx_pass_stats": (ctypes.c_int, [_P, ctypes.POINTER(FhxStats)]),
    "fhx_set_global_stats": (ctypes.c_int, [_P, ctypes.POINTER(FhxStats), _I64P, _I64P, ctypes.c_int64]),
    "fhx_set_dist_keys": (ctypes.c_int, [_P, _I64P, ctypes.c_int64]),
    "fhx_set_outlier_dists": (ctypes.c_int, [_P, _I64P, ctypes.c_int64]),
    "fhx_set_outlier_dist_hist": (ctypes.c_int, [_P, _I64P, ctypes.c_int64]),
    "fhx_make_bins": (ctypes.c_int, [_P, _I32P]),
    "fhx_fit": (ctypes.c_int, [_P, ctypes.POINTER(FhxFitInfo)]),
    "fhx_pvalues": (ctypes.c_int, [_P]),
    "fhx_bh": (ctypes.c_int, [_P, ctypes.c_double]),
    "fhx_sync": (ctypes.c_int, [_P]),
    "fhx_set_global_rows": (ctypes.c_int, [_P, _I64P, ctypes.c_int64]),
    "fhx_get_skip_limit": (ctypes.c_int64, [_P]),
    "fhx_set_skip_limit": (ctypes.c_int, [_P, ctypes.c_int64]),
    "fhx_next_pass": (ctypes.c_int, [_P, _I64P]),
    "fhx_reset_passes": (ctypes.c_int, [_P]),
    "fhx_get_stats": (ctypes.c_int, [_P, ctypes.POINTER(FhxStats)]),
    "fhx_fetch": (ctypes.c_int, [_P, _F64P, _F64P, _F64P, _F64P, _F64P]),
    "fhx_fetch_flags": (ctypes.c_int, [_P, ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_uint8)]),
    "fhx_fetch_outlier_rows": (ctypes.c_int, [_P, ctypes.POINTER(ctypes.c_int64), ctypes.c_int64, ctypes.POINTER(ctypes.c_int64)]),
    "fhx_get_array": (ctypes.c_int, [_P, ctypes.c_int, _P, ctypes.c_int64, _I64P]),
    "fhx_device_ptr": (_P, [_P, ctypes.c_int]),
    "fhx_n_sorted": (ctypes.c_int64, [_P]),
    "fhx_bh_sort_stats": (ctypes.c_int, [_P, _I64P]),
    "fhx_kernel_seconds": (ctypes.c_int, [_P, _F64P, _F64P, _F64P]),
    "fhx_kernel_seconds_total": (ctypes.c_int, [_P, _F64P, _I64P, ctypes.c_int]),
    "fhx_kernel_events_dropped": (ctypes.c_int, [_P, _I64P]),
    "fhx_k2_heavy_launch": (ctypes.c_int, [_P, _F64P, _I64P]),
    "fhx_k2_heavy_clock": (ctypes.c_int, [_P, _F64P]),
    "fhx_k2_class_rows": (ctypes.c_int, [_P, _I64P]),
    "fhx_bdtrc_array": (ctypes.c_int, [_P, ctypes.c_double, _I32P, _F64P, ctypes.c_int64, _F64P]),
    "fhx_debug_format": (ctypes.c_int, [_P, _F64P, ctypes.c_int64, ctypes.c_int32, ctypes.c_char_p, _I32P]),
    "fhx_debug_contfrac": (ctypes.c_int, [_P, ctypes.c_int, ctypes.c_int, _F64P, _F64P, _F64P, ctypes.c_int64, _F64P]),
    "fhx_debug_lean_div": (ctypes.c_int, [_P, _F64P, _F64P, ctypes.c_int64, _F64P]),
    "fhx_debug_classify": (ctypes.c_int, [_P, ctypes.c_double, _I32P, _F64P, ctypes.c_int64, _I32P, _I32P, _F64P]),
    "fhx_bh_array": (ctypes.c_int, [_P, _F64P, ctypes.c_int64, ctypes.c_double, _F64P]),
    "fhx_bh_top_hist": (ctypes.c_int, [_P, _I64P, ctypes.c_int64]),
    "fhx_bh_set_cutoff": (ctypes.c_int, [_P, _I64P, ctypes.c_int64, ctypes.c_double]),
    "fhx_bh_top_hist_device": (ctypes.c_int, [_P]),
    "fhx_bh_set_cutoff_device": (ctypes.c_int, [_P, ctypes.c_double]),
    "fhx_bh_local_sort": (ctypes.c_int, [_P]),
    "fhx_bh_apply_sorted": (ctypes.c_int, [_P, _P, ctypes.c_int64, ctypes.c_int64, ctypes.c_double, ctypes.c_double, _P, _F64P]),
    "fhx_sort_u64": (ctypes.c_int, [_P, _P, ctypes.c_int64, _P, _P]),
    "fhx_bh_scatter": (ctypes.c_int, [_P, _P]),
    "fhx_memcpy_d2d": (ctypes.c_int, [_P, _P, _P, ctypes.c_int64]),
    "fhx_comm_unique_id": (ctypes.c_int, [_P, ctypes.c_int64]),
    "fhx_comm_init": (ctypes.c_int, [_P, _P, ctypes.c_int, ctypes.c_int]),
    "fhx_comm_init_custom": (ctypes.c_int, [_P, _P, ctypes.c_int, ctypes.c_int]),
    "fhx_comm_destroy": (ctypes.c_int, [_P]),
    "fhx_comm_info": (ctypes.c_int, [_P, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "fhx_run_pass_distributed": (ctypes.c_int, [_P, ctypes.POINTER(FhxFitInfo)]),
    "fhx_next_pass_distributed": (ctypes.c_int, [_P, _I64P]),
    "fhx_pass_stats_distributed": (ctypes.c_int, [_P, ctypes.POINTER(FhxStats)]),
    "fhx_bh_distributed": (ctypes.c_int, [_P, ctypes.c_double]),
    "fhx_dist_stage_seconds": (ctypes.c_int, [_P, _F64P]),
    "fhx_dist_trace": (ctypes.c_int, [_P, ctypes.c_char_p, ctypes.c_int64, _I64P, ctypes.c_int]),
    "fhx_copy": (ctypes.c_int, [_P, _P, _P, ctypes.c_int64, ctypes.c_int]),
    "fhx_host_read_table": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(_P)]),
    "fhx_table_rows": (ctypes.c_int64, [_P]),
    "fhx_table_n_names": (ctypes.c_int32, [_P]),
    "fhx_table_name": (ctypes.c_char_p, [_P, ctypes.c_int32]),
    "fhx_table_error": (ctypes.c_char_p, [_P]),
    "fhx_table_copy": (ctypes.c_int, [_P, ctypes.c_int32, _P]),
    "fhx_table_map_names": (ctypes.c_int, [_P, _I32P, ctypes.c_int32]),
    "fhx_table_free": (None, [_P]),
    "fhx_host_write_contacts": (ctypes.c_int, [ctypes.c_char_p, ctypes.POINTER(ctypes.c_char_p), ctypes.c_int32, _I32P, _I32P, _I32P, _I32P,
                                               _I32P, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32]),
    "fhx_host_write_significances": (ctypes.c_int, [ctypes.c_char_p, ctypes.POINTER(ctypes.c_char_p), ctypes.c_int32, _I32P, _I32P, _I32P,
                                                    _I32P, _I32P, _F64P, _F64P, _F64P, _F64P, _F64P, ctypes.c_int64, ctypes.c_int32,
                                                    ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, _I64P]),
    "fhx_write_significances_device": (ctypes.c_int, [_P, ctypes.c_char_p, ctypes.POINTER(ctypes.c_char_p), ctypes.c_int32, _I32P, _I32P,
                                                      _I32P, _I32P, _I32P, ctypes.c_int64, _I64P, _I64P]),
    "fhx_write_significances_device_range": (ctypes.c_int, [_P, ctypes.c_char_p, ctypes.POINTER(ctypes.c_char_p), ctypes.c_int32, ctypes.c_int64,
                                                            ctypes.c_int64, ctypes.c_int32, _I64P, _I64P]),
    "fhx_ingest_contacts_file_slice": (ctypes.c_int, [_P, ctypes.c_char_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _I64P, _I32P, _I32P, _I32P]),
    "fhx_set_global_rows_range": (ctypes.c_int, [_P, ctypes.c_int64]),
    "fhx_text_part_bounds": (ctypes.c_int, [_P, ctypes.c_int32, ctypes.c_int32, _I64P, _I64P]),
    "fhx_ingest_contacts_text_slice": (ctypes.c_int, [_P, _P, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _I64P, _I32P]),
    "fhx_host_inflate_part": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(_P)]),
    "fhx_text_part_bytes": (ctypes.c_int64, [_P]),
    "fhx_text_part_is_last": (ctypes.c_int32, [_P]),
    "fhx_text_part_tail": (ctypes.c_int, [_P, _P, ctypes.c_int64]),
    "fhx_text_part_resolve": (ctypes.c_int, [_P, _P, ctypes.c_int64, ctypes.POINTER(_P), ctypes.POINTER(ctypes.c_uint32)]),
    "fhx_text_part_error": (ctypes.c_char_p, [_P]),
    "fhx_text_part_free": (None, [_P]),
    "fhx_crc32_combine": (ctypes.c_uint32, [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int64]),
    "fhx_text_first_row_end": (ctypes.c_int64, [_P, _P, ctypes.c_int64]),
    "fhx_text_ends_with_newline": (ctypes.c_int32, [_P]),
    "fhx_ingest_contacts_text_own": (ctypes.c_int, [_P, _P, ctypes.c_int32, ctypes.c_int64, ctypes.c_char_p, ctypes.c_int64, _I64P, _I32P]),
    "fhx_ingest_contacts_chr_counts": (ctypes.c_int, [_P, _I64P, ctypes.c_int32]),
    "fhx_ingest_contacts_commit_shard": (ctypes.c_int, [_P, _I32P, ctypes.POINTER(ctypes.c_uint8), ctypes.c_int32, _I64P]),
    "fhx_shard_segments": (ctypes.c_int, [_P, _I64P, _I64P, _I64P, ctypes.c_int64, _I64P]),
    "fhx_host_spline_fit": (ctypes.c_int, [_F64P, _F64P, ctypes.c_int32, ctypes.c_double, _F64P, _F64P, _I32P, _F64P, _I32P, _I32P]),
    "fhx_host_spline_eval": (ctypes.c_int, [_F64P, _F64P, ctypes.c_int32, _F64P, ctypes.c_int64, _F64P]),
    "fhx_host_pava_decreasing": (ctypes.c_int, [_F64P, ctypes.c_int64, _F64P]),
    "fhx_host_lbeta_table": (ctypes.c_int, [ctypes.c_double, ctypes.c_int64, _F64P, _F64P]),
    # Knight-Ruiz bias vectors (fithic/utils/HiCKRy.py)
    "fhx_kr_create": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    "fhx_kr_destroy": (None, [ctypes.c_void_p]),
    "fhx_kr_last_error": (ctypes.c_char_p, [ctypes.c_void_p]),
    "fhx_kr_load_loci": (ctypes.c_int, [ctypes.c_void_p, _I32P, _I32P, ctypes.c_int64]),
    "fhx_kr_load_pairs": (ctypes.c_int, [ctypes.c_void_p, _I32P, _I32P, _I32P, _I32P, _F64P, ctypes.c_int64, _I64P]),
    "fhx_kr_shape": (ctypes.c_int, [ctypes.c_void_p, _I64P, _I64P, _I64P, _I64P]),
    "fhx_kr_get_csr": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, _I64P, _I32P, _F64P]),
    "fhx_kr_row_sums": (ctypes.c_int, [ctypes.c_void_p, _F64P]),
    "fhx_kr_remove_sparse": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_double, _I64P, _F64P, _I64P]),
    "fhx_kr_get_removed": (ctypes.c_int, [ctypes.c_void_p, _I64P, ctypes.c_int64, _I64P]),
    "fhx_kr_balance": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p]),
    "fhx_kr_get_x": (ctypes.c_int, [ctypes.c_void_p, _F64P]),
    "fhx_kr_bias": (ctypes.c_int, [ctypes.c_void_p, _F64P]),
    "fhx_kr_spmv": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, _F64P, _F64P, ctypes.c_int32, _F64P]),
    "fhx_kr_dot": (ctypes.c_int, [ctypes.c_void_p, _F64P, _F64P, ctypes.c_int64, _F64P]),
    # merging of nearby significant contacts (fithic/utils/CombineNearbyInteraction.py)
    "fhx_cni_create": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    "fhx_cni_destroy": (None, [ctypes.c_void_p]),
    "fhx_cni_last_error": (ctypes.c_char_p, [ctypes.c_void_p]),
    "fhx_cni_load": (ctypes.c_int, [ctypes.c_void_p, _I32P, _I64P, _I64P, _I64P, _F64P, _F64P, ctypes.c_int64, ctypes.c_int64, _I64P]),
    "fhx_cni_run": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]),
    "fhx_cni_get_records": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, _I64P]),
}

# One object per translation unit (fithic_amd/csrc/_obj/, git-ignored), compiled in parallel, then one link: a change in K2 does
# not recompile K1, K3, the Knight-Ruiz path or the host stages.  Flags are the same for every unit - -ffp-contract=off matters
# for bit-exactness on the host (FITPACK, lgamma tables) as much as on the device (fhx_bdtrc.hpp).
SOURCES = ["fhx_device.hip", "fhx_k1.hip", "fhx_k2.hip", "fhx_k3.hip", "fhx_kr.hip", "fhx_cni.hip", "fhx_host.cpp", "fhx_io.cpp", "fhx_gunzip.cpp"]
COMPILE_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-pthread"]
LINK_FLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-lz", "-ldl"]
OBJ_DIR = os.path.join(CSRC, "_obj")


def build_commands():
    """[(compile command per unit)], link command - what build() runs (INTEGRATION.md quotes it)"""
    objs = [os.path.join(OBJ_DIR, os.path.splitext(f)[0] + ".o") for f in SOURCES]
    compiles = [["hipcc"] + COMPILE_FLAGS + ["-c", "-o", o, os.path.join(CSRC, f)] for f, o in zip(SOURCES, objs)]
    link = ["hipcc"] + objs + LINK_FLAGS + ["-o", LIB_PATH]
    return compiles, link


def build(force=False):
    """Compile the HIP kernels + host stages for gfx950 into the in-tree shared library."""
    from concurrent.futures import ThreadPoolExecutor
    shared = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".inc", ".def", ".h"))]
    header = os.path.join(os.path.dirname(_PKG), "include", "fithic_mi355x.h")
    if os.path.exists(header):                         # an installed package ships csrc/ but not include/
        shared.append(header)
    newest_shared = max(os.path.getmtime(p) for p in shared)
    compiles, link = build_commands()
    # a library newer than every source and header is current whatever csrc/_obj holds: the objects are git- and gpurun-ignored,
    # so a fresh checkout or the GPU box has the .so without them and must not recompile nine units to find that out
    newest_src = max([newest_shared] + [os.path.getmtime(os.path.join(CSRC, f)) for f in SOURCES])
    if not force and os.path.exists(LIB_PATH) and os.path.getmtime(LIB_PATH) >= newest_src:
        return LIB_PATH
    os.makedirs(OBJ_DIR, exist_ok=True)
    stale = []
    for f, cmd in zip(SOURCES, compiles):
        obj, src = cmd[-2], cmd[-1]
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), newest_shared):
            stale.append(cmd)
    if stale:
        def run(cmd):
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("%s failed:\n%s" % (" ".join(cmd[-4:]), r.stderr[-4000:]))
        with ThreadPoolExecutor(max_workers=min(len(stale), os.cpu_count() or 4)) as ex:
            list(ex.map(run, stale))
    objs = [cmd[-2] for cmd in compiles]
    if stale or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(o) for o in objs):
        subprocess.check_call(link)
    return LIB_PATH


_lib = None


def _share_torch_rccl():
    """One RCCL per process, for the same reason: PyTorch ships a librccl.so built against its own HIP runtime.  When torch is
    installed its copy is loaded (globally) before the first communicator is made and the library's run-time lookup finds it;
    without torch the system's librccl.so.1 is used."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return None
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "librccl.so")
    if not os.path.exists(cand):
        return None
    try:
        return ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
    except OSError:
        return None


UNIQUE_ID_BYTES = 128
STAGE_NAMES = ("k1_and_stats_exchange", "host_fit", "k2_launch", "cutoff_sort_splitters", "exchange_slice_bh_return")


def comm_unique_id():
    """ncclGetUniqueId as 128 bytes (rank 0 makes it, every rank passes it to Context.comm_init)."""
    _share_torch_rccl()
    buf = ctypes.create_string_buffer(UNIQUE_ID_BYTES)
    rc = lib().fhx_comm_unique_id(ctypes.cast(buf, _P), UNIQUE_ID_BYTES)
    if rc != FHX_OK:
        raise FhxError(rc, "fhx_comm_unique_id: RCCL is not available")
    return bytes(buf.raw)


class FhxTransport(ctypes.Structure):
    """Caller-provided collectives on device pointers (include/fithic_mi355x.h: fhx_transport)."""
    ALL_REDUCE = ctypes.CFUNCTYPE(ctypes.c_int, _P, _P, ctypes.c_int64, ctypes.c_int)
    ALL_GATHER = ctypes.CFUNCTYPE(ctypes.c_int, _P, _P, _P, ctypes.c_int64)
    ALL_TO_ALL_V = ctypes.CFUNCTYPE(ctypes.c_int, _P, _P, _I64P, _I64P, _P, _I64P, _I64P, ctypes.c_int)
    _fields_ = [("user", _P), ("all_reduce_i64", ALL_REDUCE), ("all_gather", ALL_GATHER), ("all_to_all_v", ALL_TO_ALL_V)]


def lib():
    """The loaded library; raises (loudly) when it has not been built - there is no fallback."""
    global _lib
    if _lib is None:
        L = _loader.load()
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)        # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


class FhxError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("fhx error %d: %s" % (code, message))
        self.code = code


def _ptr(a, ctype):
    return a.ctypes.data_as(ctypes.POINTER(ctype))


def _i32(a):
    a = np.ascontiguousarray(a)
    if a.dtype != np.int32:
        if a.size and (a.min() < -(1 << 31) or a.max() >= (1 << 31)):
            raise ValueError("value does not fit int32")
        a = a.astype(np.int32)
    return a


class Context:
    """One engine context = one GPU (device >= 0) or host-only (device = -1)."""

    def __init__(self, device=0):
        self._L = lib()
        h = _P()
        rc = self._L.fhx_create(int(device), ctypes.byref(h))
        if rc != FHX_OK:
            raise FhxError(rc, "fhx_create(device=%d) failed%s" % (device, " - no usable MI355X" if rc == FHX_ERR_NO_DEVICE else ""))
        self._h = h
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            self._L.fhx_destroy(self._h)
            self._h = None

    __del__ = close

    def _check(self, rc):
        if rc != FHX_OK:
            raise FhxError(rc, (self._L.fhx_last_error(self._h) or b"").decode())

    # ---- setup ----
    def set_params(self, resolution, dist_low=0, dist_up=None, n_bins=100, mapp_thres=1, mode=MODE_INTRA_ONLY,
                   bias_low=0.5, bias_up=2.0, totals=TOTALS_REFERENCE):
        up = INT64_MAX if dist_up is None or dist_up == float("inf") else int(dist_up)
        p = FhxParams(int(resolution), int(dist_low), up, int(n_bins), int(mapp_thres), int(mode), int(totals), float(bias_low),
                      float(bias_up))
        self._check(self._L.fhx_set_params(self._h, ctypes.byref(p)))

    def load_fragments(self, chr_ids, mids, hits, chr_sort_rank):
        c, m, h, r = _i32(chr_ids), _i32(mids), _i32(hits), _i32(chr_sort_rank)
        self._check(self._L.fhx_load_fragments(self._h, _ptr(c, ctypes.c_int32), _ptr(m, ctypes.c_int32), _ptr(h, ctypes.c_int32),
                                               len(c), _ptr(r, ctypes.c_int32), len(r)))

    def load_bias(self, chr_ids, mids, bias):
        c, m = _i32(chr_ids), _i32(mids)
        b = np.ascontiguousarray(bias, np.float64)
        self._check(self._L.fhx_load_bias(self._h, _ptr(c, ctypes.c_int32), _ptr(m, ctypes.c_int32), _ptr(b, ctypes.c_double), len(c)))

    def load_pairs(self, chr1, mid1, chr2, mid2, count):
        a = [_i32(v) for v in (chr1, mid1, chr2, mid2, count)]
        self._check(self._L.fhx_load_pairs(self._h, *[_ptr(v, ctypes.c_int32) for v in a], len(a[0])))

    def load_pairs_device(self, ptrs, n, stream=None):
        self._check(self._L.fhx_load_pairs_device(self._h, *[_P(int(p)) for p in ptrs], int(n), _P(stream or 0)))

    def ingest_contacts_text(self, text, threads=0):
        """The inflated contacts file (host_inflate) parsed on the GPU -> (rows, names in order of first appearance).  Raises
        FhxError(FHX_ERR_UNSUPPORTED) for a file the device parser does not take: host_parse_text(text, ...) then."""
        n, k = ctypes.c_int64(0), ctypes.c_int32(0)
        self._check(self._L.fhx_ingest_contacts_text(self._h, text._h, int(threads), ctypes.byref(n), ctypes.byref(k)))
        return n.value, [self._L.fhx_ingest_contacts_name(self._h, i).decode() for i in range(k.value)]

    def ingest_contacts_file(self, path, threads=0):
        """The contacts FILE inflated and parsed on the GPU -> (rows, names).  Raises FhxError(FHX_ERR_UNSUPPORTED) with
        .refused = 1 (container / stream not taken: inflate on the host, the device parser may still take the text) or
        2 (text outside the device grammar: the host parser)."""
        n, k, why = ctypes.c_int64(0), ctypes.c_int32(0), ctypes.c_int32(0)
        rc = self._L.fhx_ingest_contacts_file(self._h, os.fsencode(path), int(threads), ctypes.byref(n), ctypes.byref(k), ctypes.byref(why))
        if rc != FHX_OK:
            e = FhxError(rc, (self._L.fhx_last_error(self._h) or b"").decode())
            e.refused = why.value
            raise e
        return n.value, [self._L.fhx_ingest_contacts_name(self._h, i).decode() for i in range(k.value)]

    def debug_inflate_file(self, path, cap):
        out = np.empty(int(cap), np.uint8)
        n = ctypes.c_int64(0)
        self._check(self._L.fhx_debug_inflate_file(self._h, os.fsencode(path), out.ctypes.data_as(_P), int(cap), ctypes.byref(n)))
        return out[:n.value].tobytes()

    def ingest_contacts_commit(self, ids):
        """ids[i] = the run's chromosome id of name i: the parsed rows become the context's contact rows"""
        ids = np.ascontiguousarray(ids, np.int32)
        self._check(self._L.fhx_ingest_contacts_commit(self._h, _ptr(ids, ctypes.c_int32), len(ids)))

    def ingest_contacts_file_slice(self, path, part, n_parts, threads=0):
        """part `part` of `n_parts` of the contacts FILE inflated and parsed on the GPU -> (rows, names, ends with a newline); raises
        FhxError(FHX_ERR_UNSUPPORTED) with .refused as ingest_contacts_file"""
        n, k, why, nl = ctypes.c_int64(0), ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int32(0)
        rc = self._L.fhx_ingest_contacts_file_slice(self._h, os.fsencode(path), int(threads), int(part), int(n_parts), ctypes.byref(n),
                                                    ctypes.byref(k), ctypes.byref(why), ctypes.byref(nl))
        if rc != FHX_OK:
            e = FhxError(rc, (self._L.fhx_last_error(self._h) or b"").decode())
            e.refused = why.value
            raise e
        return n.value, [self._L.fhx_ingest_contacts_name(self._h, i).decode() for i in range(k.value)], bool(nl.value)

    def ingest_contacts_text_slice(self, text, part, n_parts, threads=0):
        """the rows that start in part `part` of `n_parts` of an inflated text (HostText) parsed on the GPU -> (rows, names)"""
        n, k = ctypes.c_int64(0), ctypes.c_int32(0)
        self._check(self._L.fhx_ingest_contacts_text_slice(self._h, text._h, int(threads), int(part), int(n_parts), ctypes.byref(n), ctypes.byref(k)))
        return n.value, [self._L.fhx_ingest_contacts_name(self._h, i).decode() for i in range(k.value)]

    def ingest_contacts_text_own(self, text, skip, extra=b"", threads=0):
        """the rows whose first byte lies in `text` (a part of a stream: HostText of TextPart.resolve) from byte `skip` on, the last one
        completed by `extra`, parsed on the GPU -> (rows, names)"""
        n, k = ctypes.c_int64(0), ctypes.c_int32(0)
        self._keep_extra = bytes(extra)                       # (referenced until the call returns)
        self._check(self._L.fhx_ingest_contacts_text_own(self._h, text._h, int(threads), int(skip), self._keep_extra, len(self._keep_extra),
                                                         ctypes.byref(n), ctypes.byref(k)))
        return n.value, [self._L.fhx_ingest_contacts_name(self._h, i).decode() for i in range(k.value)]

    def set_global_rows_range(self, first):
        self._check(self._L.fhx_set_global_rows_range(self._h, int(first)))

    def ingest_contacts_chr_counts(self, n_names):
        """rows of the parsed text per name, by the chromosome of the first locus"""
        out = np.zeros(int(n_names), np.int64)
        self._check(self._L.fhx_ingest_contacts_chr_counts(self._h, _ptr(out, ctypes.c_int64), int(n_names)))
        return out

    def ingest_contacts_commit_shard(self, ids, mine):
        """as ingest_contacts_commit, keeping only the rows whose first chromosome has mine[i] set -> rows kept"""
        ids = np.ascontiguousarray(ids, np.int32)
        mine = np.ascontiguousarray(mine, np.uint8)
        n = ctypes.c_int64(0)
        self._check(self._L.fhx_ingest_contacts_commit_shard(self._h, _ptr(ids, ctypes.c_int32), _ptr(mine, ctypes.c_uint8), len(ids), ctypes.byref(n)))
        return n.value

    def shard_segments(self, cap=1 << 10):
        """[(local start, file position, length)] of the stretches of consecutive file positions this rank holds, or None (> cap)"""
        a, b, c = (np.zeros(int(cap), np.int64) for _ in range(3))
        n = ctypes.c_int64(0)
        self._check(self._L.fhx_shard_segments(self._h, _ptr(a, ctypes.c_int64), _ptr(b, ctypes.c_int64), _ptr(c, ctypes.c_int64), int(cap), ctypes.byref(n)))
        if n.value > cap:
            return None
        return [(int(a[i]), int(b[i]), int(c[i])) for i in range(n.value)]

    def write_significances_range(self, path, chr_names, row_begin, row_end, with_header):
        """rows [row_begin, row_end) of this context as gzip members (header member in front if asked) -> (rows written, bytes)"""
        arr = (ctypes.c_char_p * len(chr_names))(*[os.fsencode(n) for n in chr_names])
        rows, nbytes = ctypes.c_int64(0), ctypes.c_int64(0)
        self._check(self._L.fhx_write_significances_device_range(self._h, os.fsencode(path), arr, len(chr_names), int(row_begin), int(row_end),
                                                                 1 if with_header else 0, ctypes.byref(rows), ctypes.byref(nbytes)))
        return rows.value, nbytes.value

    def ingest_contacts_discard(self):
        self._L.fhx_ingest_contacts_discard(self._h)

    def fetch_pairs(self, rows=None, n=None):
        """(chr1, mid1, chr2, mid2, count) of the given loaded rows (all n of them when rows is None), rebuilt on the device"""
        if rows is not None:
            rows = np.ascontiguousarray(rows, np.int64)
            n = len(rows)
        out = [np.empty(int(n), np.int32) for _ in range(5)]
        self._check(self._L.fhx_fetch_pairs(self._h, _ptr(rows, ctypes.c_int64) if rows is not None else None, int(n),
                                            *[_ptr(v, ctypes.c_int32) for v in out]))
        return out

    # ---- one pass ----
    def pass_stats(self):
        st = FhxStats()
        self._check(self._L.fhx_pass_stats(self._h, ctypes.byref(st)))
        return st

    def set_global_stats(self, stats, hist_sumcc, hist_npairs):
        a = np.ascontiguousarray(hist_sumcc, np.int64)
        b = np.ascontiguousarray(hist_npairs, np.int64)
        self._check(self._L.fhx_set_global_stats(self._h, ctypes.byref(stats), _ptr(a, ctypes.c_int64), _ptr(b, ctypes.c_int64), len(a)))

    def set_dist_keys(self, keys):
        a = np.ascontiguousarray(keys, np.int64)
        self._check(self._L.fhx_set_dist_keys(self._h, _ptr(a, ctypes.c_int64), len(a)))

    def set_outlier_dists(self, dists):
        a = np.ascontiguousarray(dists, np.int64)
        self._check(self._L.fhx_set_outlier_dists(self._h, _ptr(a, ctypes.c_int64), len(a)))

    def set_outlier_dist_hist(self, hist):
        a = np.ascontiguousarray(hist, np.int64)
        self._check(self._L.fhx_set_outlier_dist_hist(self._h, _ptr(a, ctypes.c_int64), len(a)))

    def make_bins(self):
        n = ctypes.c_int32(0)
        self._check(self._L.fhx_make_bins(self._h, ctypes.byref(n)))
        return n.value

    def fit(self):
        info = FhxFitInfo()
        self._check(self._L.fhx_fit(self._h, ctypes.byref(info)))
        return info

    def run_pass(self):
        """pass_stats -> fit -> pvalues -> bh in one call (fhx_run_pass) -> (FhxStats, FhxFitInfo)"""
        st, info = FhxStats(), FhxFitInfo()
        self._check(self._L.fhx_run_pass(self._h, ctypes.byref(st), ctypes.byref(info)))
        return st, info

    def pvalues(self):
        self._check(self._L.fhx_pvalues(self._h))

    def bh(self, n_total_tests):
        self._check(self._L.fhx_bh(self._h, float(n_total_tests)))

    def sync(self):
        self._check(self._L.fhx_sync(self._h))

    def set_global_rows(self, rows):
        r = np.ascontiguousarray(rows, np.int64)
        self._check(self._L.fhx_set_global_rows(self._h, _ptr(r, ctypes.c_int64), len(r)))

    def get_skip_limit(self):
        return self._L.fhx_get_skip_limit(self._h)

    def set_skip_limit(self, limit):
        self._check(self._L.fhx_set_skip_limit(self._h, int(limit)))

    def next_pass(self):
        n = ctypes.c_int64(0)
        self._check(self._L.fhx_next_pass(self._h, ctypes.byref(n)))
        return n.value

    def reset_passes(self):
        self._check(self._L.fhx_reset_passes(self._h))

    # ---- sharded runs (one context per GPU / rank) ----
    def comm_init(self, unique_id, rank, world):
        """RCCL communicator on this context's GPU (blocks until every rank has called it)."""
        _share_torch_rccl()
        buf = ctypes.create_string_buffer(bytes(unique_id), UNIQUE_ID_BYTES)
        self._check(self._L.fhx_comm_init(self._h, ctypes.cast(buf, _P), int(rank), int(world)))

    def comm_init_custom(self, transport, rank, world):
        self._transport = transport                      # keeps the callbacks alive
        self._check(self._L.fhx_comm_init_custom(self._h, ctypes.cast(ctypes.pointer(transport), _P), int(rank), int(world)))

    def comm_destroy(self):
        self._check(self._L.fhx_comm_destroy(self._h))

    def comm_info(self):
        r, w, v = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        self._check(self._L.fhx_comm_info(self._h, ctypes.byref(r), ctypes.byref(w), ctypes.byref(v)))
        return r.value, w.value, v.value

    def run_pass_distributed(self):
        info = FhxFitInfo()
        self._check(self._L.fhx_run_pass_distributed(self._h, ctypes.byref(info)))
        return info

    def pass_stats_distributed(self):
        st = FhxStats()
        self._check(self._L.fhx_pass_stats_distributed(self._h, ctypes.byref(st)))
        return st

    def bh_distributed(self, n_total_tests):
        self._check(self._L.fhx_bh_distributed(self._h, float(n_total_tests)))

    def next_pass_distributed(self):
        n = ctypes.c_int64(0)
        self._check(self._L.fhx_next_pass_distributed(self._h, ctypes.byref(n)))
        return n.value

    def dist_stage_seconds(self):
        out = np.zeros(5, np.float64)
        self._check(self._L.fhx_dist_stage_seconds(self._h, _ptr(out, ctypes.c_double)))
        return dict(zip(STAGE_NAMES, out.tolist()))

    def dist_trace(self, clear=True):
        """[(step, kind, size)] of the collectives issued so far (FHX_DIST_TRACE=1 at comm_init time), in the vocabulary of
        csrc/fhx_dist_schedule.def"""
        n = ctypes.c_int64(0)
        self._check(self._L.fhx_dist_trace(self._h, None, 0, ctypes.byref(n), 0))
        buf = ctypes.create_string_buffer(max(int(n.value), 1))
        self._check(self._L.fhx_dist_trace(self._h, buf, int(n.value), ctypes.byref(n), 1 if clear else 0))
        rows = []
        for line in buf.raw[:n.value].decode().splitlines():
            step, kind, size = line.split()
            rows.append((step, kind, int(size)))
        return rows

    def copy(self, dst, src, nbytes, kind):
        """kind 0 = host to device, 1 = device to host, 2 = device to device (addresses as ints)."""
        self._check(self._L.fhx_copy(self._h, _P(int(dst)), _P(int(src)), int(nbytes), int(kind)))

    def stats(self):
        """fhx_stats of the last pass as the context holds them (global ones after a sharded pass)."""
        st = FhxStats()
        self._check(self._L.fhx_get_stats(self._h, ctypes.byref(st)))
        return st

    def fetch(self, n_rows, p=True, q=True, expcc=False, bias=False):
        out = {}
        bufs = []
        for key, want in (("p", p), ("q", q), ("expcc", expcc), ("b1", bias), ("b2", bias)):
            if want:
                out[key] = np.empty(n_rows, np.float64)
                bufs.append(_ptr(out[key], ctypes.c_double))
            else:
                bufs.append(None)
        self._check(self._L.fhx_fetch(self._h, *bufs))
        return out

    def fetch_outlier_rows(self):
        """row numbers of the outlier lines, ascending (int64), compacted on the device"""
        n = ctypes.c_int64(0)
        self._check(self._L.fhx_fetch_outlier_rows(self._h, None, 0, ctypes.byref(n)))
        rows = np.empty(n.value, np.int64)
        if n.value:
            self._check(self._L.fhx_fetch_outlier_rows(self._h, _ptr(rows, ctypes.c_int64), len(rows), ctypes.byref(n)))
        return rows

    def fetch_flags(self, n_rows, outlier=True, skip=False):
        o = np.empty(n_rows, np.uint8) if outlier else None
        k = np.empty(n_rows, np.uint8) if skip else None
        self._check(self._L.fhx_fetch_flags(self._h, _ptr(o, ctypes.c_uint8) if outlier else None,
                                            _ptr(k, ctypes.c_uint8) if skip else None))
        return o, k

    def write_significances_device(self, path, names, chr1=None, mid1=None, chr2=None, mid2=None, count=None):
        """The significances file formatted and deflated by the GPU from the resident p and q; returns (rows, bytes).  Without
        the five identity columns they are rebuilt on the device from the rows fhx_load_pairs stored (nothing is uploaded).
        Raises FhxError with code FHX_ERR_UNSUPPORTED when a row does not fit the device formatter (use host_write_significances)."""
        arr_names = (ctypes.c_char_p * len(names))(*[n.encode() for n in names])
        rows, nbytes = ctypes.c_int64(0), ctypes.c_int64(0)
        if chr1 is None:
            ptrs, n = [None] * 5, self.stats().n_rows
        else:
            i32 = [_i32(v) for v in (chr1, mid1, chr2, mid2, count)]
            ptrs, n = [_ptr(v, ctypes.c_int32) for v in i32], len(i32[0])
        self._check(self._L.fhx_write_significances_device(self._h, os.fsencode(path), arr_names, len(names), *ptrs, n, ctypes.byref(rows),
                                                           ctypes.byref(nbytes)))
        return rows.value, nbytes.value

    def get_array(self, which):
        n = ctypes.c_int64(0)
        self._check(self._L.fhx_get_array(self._h, which, None, 0, ctypes.byref(n)))
        a = np.empty(n.value, _ARRAY_DTYPE.get(which, np.int64))
        self._check(self._L.fhx_get_array(self._h, which, a.ctypes.data_as(_P), n.value, ctypes.byref(n)))
        return a

    def kernel_seconds(self):
        k = [ctypes.c_double(0) for _ in range(3)]
        self._check(self._L.fhx_kernel_seconds(self._h, *[ctypes.byref(v) for v in k]))
        return tuple(v.value for v in k)

    def kernel_seconds_total(self, reset=False):
        """(seconds[4], passes[4]) of K1, K2, K3 and the heavy K2 launch summed since the last reset (waits for the stream once)."""
        s = (ctypes.c_double * 4)()
        n = (ctypes.c_int64 * 4)()
        self._check(self._L.fhx_kernel_seconds_total(self._h, s, n, 1 if reset else 0))
        return list(s), list(n)

    def bh_sort_stats(self):
        """how the last BH call sorted its survivors (include/fithic_mi355x.h: fhx_bh_sort_stats)"""
        n = (ctypes.c_int64 * 8)()
        self._check(self._L.fhx_bh_sort_stats(self._h, n))
        return dict(zip(("passes", "low_bit", "beyond_lists", "inversions", "runs_by_thread", "long_run_inversions", "segments", "segment_keys"),
                        [int(v) for v in n]))

    def kernel_events_dropped(self):
        """passes of K1, K2, K3, heavy launch whose events were overwritten unread since the last reset (0 = the sums are complete)"""
        n = (ctypes.c_int64 * 4)()
        self._check(self._L.fhx_kernel_events_dropped(self._h, n))
        return list(n)

    def k2_heavy_launch(self):
        sec, rows = ctypes.c_double(0), ctypes.c_int64(0)
        self._check(self._L.fhx_k2_heavy_launch(self._h, ctypes.byref(sec), ctypes.byref(rows)))
        return sec.value, rows.value

    def k2_heavy_clock(self):
        """GHz the last heavy launch ran at (0.0: it held no task)"""
        ghz = ctypes.c_double(0)
        self._check(self._L.fhx_k2_heavy_clock(self._h, ctypes.byref(ghz)))
        return ghz.value

    def k2_class_rows(self):
        """rows the last pvalues() queued per class: power series, incbcf, incbd, swapped incbcf (300 iterations), closed form >= 0.01"""
        out = (ctypes.c_int64 * 5)()
        self._check(self._L.fhx_k2_class_rows(self._h, out))
        return dict(zip(("pseries", "cf_bcf", "cf_bd", "cf_swapped", "closed_pow"), [int(v) for v in out]))

    def device_ptr(self, which):
        return self._L.fhx_device_ptr(self._h, which)

    def bdtrc_array(self, n_total, count, prior):
        c = _i32(count)
        pr = np.ascontiguousarray(prior, np.float64)
        out = np.empty(len(c), np.float64)
        self._check(self._L.fhx_bdtrc_array(self._h, float(n_total), _ptr(c, ctypes.c_int32), _ptr(pr, ctypes.c_double), len(c),
                                            _ptr(out, ctypes.c_double)))
        return out

    def debug_classify(self, n_total, count, prior, thresholds=False):
        """(class by the threshold table, class by incbet's predicates[, thresholds n x 5]) for integer counts and priors"""
        c = _i32(count)
        pr = np.ascontiguousarray(prior, np.float64)
        t, a = np.empty(len(c), np.int32), np.empty(len(c), np.int32)
        thr = np.empty((len(c), 5), np.float64) if thresholds else None
        self._check(self._L.fhx_debug_classify(self._h, float(n_total), _ptr(c, ctypes.c_int32), _ptr(pr, ctypes.c_double), len(c),
                                               _ptr(t, ctypes.c_int32), _ptr(a, ctypes.c_int32),
                                               _ptr(thr, ctypes.c_double) if thresholds else None))
        return (t, a, thr) if thresholds else (t, a)

    def debug_contfrac(self, kind, lazy, a, b, x):
        a, b, x = (np.ascontiguousarray(v, np.float64) for v in (a, b, x))
        out = np.empty(len(a), np.float64)
        self._check(self._L.fhx_debug_contfrac(self._h, int(kind), int(lazy), _ptr(a, ctypes.c_double), _ptr(b, ctypes.c_double),
                                               _ptr(x, ctypes.c_double), len(a), _ptr(out, ctypes.c_double)))
        return out

    def debug_format(self, values, kind):
        """The device writer's "%e" (kind 0) / "%f" (kind 1) of every value: list of bytes, None where the device does not cover it."""
        v = np.ascontiguousarray(values, np.float64)
        text = ctypes.create_string_buffer(len(v) * 32)
        ln = np.empty(len(v), np.int32)
        self._check(self._L.fhx_debug_format(self._h, _ptr(v, ctypes.c_double), len(v), int(kind), text, _ptr(ln, ctypes.c_int32)))
        raw = text.raw
        return [raw[32 * i:32 * i + ln[i]] if ln[i] > 0 else None for i in range(len(v))]

    def debug_lean_div(self, n, d):
        n, d = (np.ascontiguousarray(v, np.float64) for v in (n, d))
        out = np.empty(len(n), np.float64)
        self._check(self._L.fhx_debug_lean_div(self._h, _ptr(n, ctypes.c_double), _ptr(d, ctypes.c_double), len(n), _ptr(out, ctypes.c_double)))
        return out

    def bh_array(self, p, n_total_tests):
        p = np.ascontiguousarray(p, np.float64)
        q = np.empty(len(p), np.float64)
        self._check(self._L.fhx_bh_array(self._h, _ptr(p, ctypes.c_double), len(p), float(n_total_tests), _ptr(q, ctypes.c_double)))
        return q

    def bh_top_hist(self):
        h = np.zeros(8192, np.int64)
        self._check(self._L.fhx_bh_top_hist(self._h, _ptr(h, ctypes.c_int64), len(h)))
        return h

    def bh_set_cutoff(self, global_hist, n_total_tests):
        h = np.ascontiguousarray(global_hist, np.int64)
        self._check(self._L.fhx_bh_set_cutoff(self._h, _ptr(h, ctypes.c_int64), len(h), float(n_total_tests)))

    def bh_top_hist_device(self):
        """Histogram left on the device; returns its address (8192 x uint64) for an in-place all-reduce."""
        self._check(self._L.fhx_bh_top_hist_device(self._h))
        return self.device_ptr(4)

    def bh_set_cutoff_device(self, n_total_tests):
        self._check(self._L.fhx_bh_set_cutoff_device(self._h, float(n_total_tests)))

    def bh_local_sort(self):
        self._check(self._L.fhx_bh_local_sort(self._h))

    def n_sorted(self):
        return self._L.fhx_n_sorted(self._h)

    def bh_apply_sorted(self, d_keys, n, rank0, carry_in, n_total_tests, d_q_sorted):
        mx = ctypes.c_double(0)
        self._check(self._L.fhx_bh_apply_sorted(self._h, _P(int(d_keys) if d_keys else 0), int(n), int(rank0), float(carry_in),
                                                float(n_total_tests), _P(int(d_q_sorted) if d_q_sorted else 0), ctypes.byref(mx)))
        return mx.value


    def sort_u64(self, d_keys_in, n, d_keys_out, d_perm_out):
        self._check(self._L.fhx_sort_u64(self._h, _P(int(d_keys_in)), int(n), _P(int(d_keys_out)), _P(int(d_perm_out))))

    def bh_scatter(self, d_q_sorted_local):
        self._check(self._L.fhx_bh_scatter(self._h, _P(int(d_q_sorted_local)) if d_q_sorted_local else None))

    def memcpy_d2d(self, dst, src, nbytes):
        self._check(self._L.fhx_memcpy_d2d(self._h, _P(int(dst)), _P(int(src)), int(nbytes)))


# ---- native text I/O (no context needed) ---------------------------------------------------------------
class HostText:
    """An inflated file (fhx_host_inflate): input of Ctx.ingest_contacts_text and of host_parse_text."""

    def __init__(self, path, threads=0):
        L = lib()
        self._L, self._h = L, _P()
        rc = L.fhx_host_inflate(os.fsencode(path), int(threads), ctypes.byref(self._h))
        if rc != FHX_OK:
            msg = (L.fhx_text_error(self._h) or b"").decode() if self._h else "fhx_host_inflate"
            self.close()
            raise FhxError(rc, msg)

    def __len__(self):
        return int(self._L.fhx_text_bytes(self._h))

    def bytes(self):
        """the inflated text (tests)"""
        out = np.empty(len(self), np.uint8)
        rc = self._L.fhx_text_copy(self._h, out.ctypes.data_as(_P), len(out))
        if rc != FHX_OK:
            raise FhxError(rc, "fhx_text_copy")
        return out.tobytes()

    def first_row_end(self):
        """(length of the first row with its newline, the row) - (-1, b"") when the text holds no newline"""
        n = int(self._L.fhx_text_first_row_end(self._h, None, 0))
        if n < 0:
            return -1, b""
        buf = ctypes.create_string_buffer(n)
        self._L.fhx_text_first_row_end(self._h, buf, n)
        return n, buf.raw

    def ends_with_newline(self):
        return bool(self._L.fhx_text_ends_with_newline(self._h))

    def part_bounds(self, part, n_parts):
        """bytes [lo, hi) of the text that part `part` of `n_parts` takes (Ctx.ingest_contacts_text_slice): the rows that start in its
        N-th of the bytes"""
        lo, hi = ctypes.c_int64(0), ctypes.c_int64(0)
        rc = self._L.fhx_text_part_bounds(self._h, int(part), int(n_parts), ctypes.byref(lo), ctypes.byref(hi))
        if rc != FHX_OK:
            raise FhxError(rc, "fhx_text_part_bounds")
        return lo.value, hi.value

    def close(self):
        if self._h:
            self._L.fhx_text_free(self._h)
            self._h = _P()

    __del__ = close


class TextPart:
    """Part `part` of `n_parts` of one plain gzip stream, decoded without the 32 KB of text before it (fhx_host_inflate_part).
    tail(): its last 32768 symbols in terms of that window; resolve(window) -> (HostText of the part, CRC-32)."""

    WINDOW = 32768

    def __init__(self, path, part, n_parts, threads=0):
        L = lib()
        self._L, self._h = L, _P()
        rc = L.fhx_host_inflate_part(os.fsencode(path), int(threads), int(part), int(n_parts), ctypes.byref(self._h))
        if rc != FHX_OK:
            msg = (L.fhx_text_part_error(self._h) or b"").decode() if self._h else "fhx_host_inflate_part"
            self.close()
            raise FhxError(rc, msg)

    def __len__(self):
        return int(self._L.fhx_text_part_bytes(self._h))

    def is_last(self):
        return bool(self._L.fhx_text_part_is_last(self._h))

    def tail(self):
        out = np.empty(self.WINDOW, np.uint16)
        rc = self._L.fhx_text_part_tail(self._h, out.ctypes.data_as(_P), len(out))
        if rc != FHX_OK:
            raise FhxError(rc, "fhx_text_part_tail")
        return out

    def resolve(self, window):
        w = np.ascontiguousarray(window, np.uint8)
        h, crc = _P(), ctypes.c_uint32(0)
        rc = self._L.fhx_text_part_resolve(self._h, w.ctypes.data_as(_P), len(w), ctypes.byref(h), ctypes.byref(crc))
        if rc != FHX_OK:
            raise FhxError(rc, "fhx_text_part_resolve")
        text = HostText.__new__(HostText)
        text._L, text._h = self._L, h
        return text, int(crc.value)

    def close(self):
        if self._h:
            self._L.fhx_text_part_free(self._h)
            self._h = _P()

    __del__ = close


def chain_windows(tails):
    """tails[r] = TextPart.tail() of part r -> the 32 KB of text before every part (part 0: zeros; nothing refers to them)"""
    windows = [np.zeros(TextPart.WINDOW, np.uint8)]
    for t in tails[:-1]:
        t = np.asarray(t, np.uint16)
        ref = t >= 256
        w = t.astype(np.uint8)
        w[ref] = windows[-1][t[ref] - 256]
        windows.append(w)
    return windows


def crc32_combine(crc_a, crc_b, len_b):
    return int(lib().fhx_crc32_combine(int(crc_a), int(crc_b), int(len_b)))


def _table_out(L, rc, h, kind, name_ids, want_float, what):
    try:
        if rc != FHX_OK:
            raise FhxError(rc, (L.fhx_table_error(h) or b"").decode() if h else what)
        n = L.fhx_table_rows(h)
        names = [L.fhx_table_name(h, i).decode() for i in range(L.fhx_table_n_names(h))]
        if name_ids is not None and names:
            ids = np.ascontiguousarray(name_ids(names), np.int32)
            rc = L.fhx_table_map_names(h, _ptr(ids, ctypes.c_int32), len(ids))
            if rc != FHX_OK:
                raise FhxError(rc, "fhx_table_map_names")
        cols = {}
        want = {0: (0, 1, 2, 3, 4), 1: (0, 1, 4), 2: (0, 1)}[kind]
        for c in want:
            a = np.empty(n, np.int32)
            L.fhx_table_copy(h, c, a.ctypes.data_as(_P))
            cols[c] = a
        dv = None
        if kind == 2 or (kind == 0 and want_float):
            dv = np.empty(n, np.float64)
            L.fhx_table_copy(h, 5, dv.ctypes.data_as(_P))
        return names, cols, dv
    finally:
        if h:
            L.fhx_table_free(h)


def host_read_table(path, kind, threads=0, name_ids=None, want_float=True):
    """-> (names, int32 columns dict, float64 column or None).  kind: 0 contacts, 1 fragments, 2 bias.
    name_ids(names) -> the caller's ids of the file's names: the chromosome columns then come out in that id space (mapped inside
    the parallel copy, not per row in numpy).  want_float=False skips the 8 B/row float column of a contacts file."""
    L = lib()
    h = _P()
    flags = 0x100 if (kind == 0 and not want_float) else 0          # FHX_TABLE_NO_FLOAT
    rc = L.fhx_host_read_table(os.fsencode(path), int(kind) | flags, int(threads), ctypes.byref(h))
    return _table_out(L, rc, h, kind, name_ids, want_float, "fhx_host_read_table")


def host_parse_text(text, kind, threads=0, name_ids=None, want_float=True):
    """host_read_table's parse stage on an already inflated file (a HostText)"""
    L = lib()
    h = _P()
    flags = 0x100 if (kind == 0 and not want_float) else 0
    rc = L.fhx_host_parse_text(text._h, int(kind) | flags, int(threads), ctypes.byref(h))
    return _table_out(L, rc, h, kind, name_ids, want_float, "fhx_host_parse_text")


def host_write_contacts(path, names, chr1, mid1, chr2, mid2, count, gzip_level=1, threads=0):
    """chr1 mid1 chr2 mid2 count as the reference reads it, size-tagged gzip members written on all cores (tooling)."""
    arr_names = (ctypes.c_char_p * len(names))(*[n.encode() for n in names])
    i32 = [_i32(v) for v in (chr1, mid1, chr2, mid2, count)]
    rc = lib().fhx_host_write_contacts(os.fsencode(path), arr_names, len(names), *[_ptr(v, ctypes.c_int32) for v in i32], len(i32[0]),
                                       int(gzip_level), int(threads))
    if rc != FHX_OK:
        raise FhxError(rc, "fhx_host_write_contacts(%s)" % path)


def host_write_significances(path, names, chr1, mid1, chr2, mid2, count, p, q, b1, b2, expcc, mode, dist_low, dist_up,
                             gzip_level=None, threads=0):
    """gzip level: 1 unless FHX_GZIP_LEVEL says otherwise - the decompressed bytes are what the reference writes; its own
    files are level 9 (Python's gzip default).  Measured on the output text (one core): level 1 compresses 0.38 M rows/s into
    39 B/row, level 3 0.28 M rows/s into 36 B/row, level 6 0.14 M rows/s into 33 B/row; formatting alone runs at 1.1 M rows/s."""
    if gzip_level is None:
        gzip_level = int(os.environ.get("FHX_GZIP_LEVEL", "1"))
    arr_names = (ctypes.c_char_p * len(names))(*[n.encode() for n in names])
    i32 = [_i32(v) for v in (chr1, mid1, chr2, mid2, count)]
    f64 = [np.ascontiguousarray(v, np.float64) for v in (p, q, b1, b2, expcc)]
    up = INT64_MAX if dist_up is None or dist_up == float("inf") else int(dist_up)
    written = ctypes.c_int64(0)
    rc = lib().fhx_host_write_significances(os.fsencode(path), arr_names, len(names), *[_ptr(v, ctypes.c_int32) for v in i32],
                                            *[_ptr(v, ctypes.c_double) for v in f64], len(i32[0]), int(mode), int(dist_low), up,
                                            int(gzip_level), int(threads), ctypes.byref(written))
    if rc != FHX_OK:
        raise FhxError(rc, "fhx_host_write_significances(%s)" % path)
    return written.value


# ---- host numerics (no context needed) -------------------------------------------------------------
def host_spline_fit(x, y, s):
    x = np.ascontiguousarray(x, np.float64)
    y = np.ascontiguousarray(y, np.float64)
    m = len(x)
    t = np.empty(m + 8, np.float64)
    c = np.empty(m + 8, np.float64)
    n = ctypes.c_int32(0)
    fp = ctypes.c_double(0)
    ier = ctypes.c_int32(0)
    rs = ctypes.c_int32(0)
    rc = lib().fhx_host_spline_fit(_ptr(x, ctypes.c_double), _ptr(y, ctypes.c_double), m, float(s), _ptr(t, ctypes.c_double),
                                   _ptr(c, ctypes.c_double), ctypes.byref(n), ctypes.byref(fp), ctypes.byref(ier), ctypes.byref(rs))
    if rc != FHX_OK:
        raise FhxError(rc, "fhx_host_spline_fit")
    return t[:n.value].copy(), c[:n.value - 4].copy(), fp.value, ier.value, bool(rs.value)


def host_spline_eval(t, c, xs):
    t = np.ascontiguousarray(t, np.float64)
    c = np.ascontiguousarray(c, np.float64)
    xs = np.ascontiguousarray(xs, np.float64)
    out = np.empty(len(xs), np.float64)
    rc = lib().fhx_host_spline_eval(_ptr(t, ctypes.c_double), _ptr(c, ctypes.c_double), len(t), _ptr(xs, ctypes.c_double), len(xs),
                                    _ptr(out, ctypes.c_double))
    if rc != FHX_OK:
        raise FhxError(rc, "fhx_host_spline_eval")
    return out


def host_pava_decreasing(y):
    y = np.ascontiguousarray(y, np.float64)
    out = np.empty(len(y), np.float64)
    rc = lib().fhx_host_pava_decreasing(_ptr(y, ctypes.c_double), len(y), _ptr(out, ctypes.c_double))
    if rc != FHX_OK:
        raise FhxError(rc, "fhx_host_pava_decreasing")
    return out


def host_lbeta_table(n_total, max_count):
    lb = np.empty(max_count + 1, np.float64)
    ib = np.empty(max_count + 1, np.float64)
    rc = lib().fhx_host_lbeta_table(float(n_total), int(max_count), _ptr(lb, ctypes.c_double), _ptr(ib, ctypes.c_double))
    if rc != FHX_OK:
        raise FhxError(rc, "fhx_host_lbeta_table")
    return lb, ib


# ---- Knight-Ruiz (fhx_kr_*) ------------------------------------------------------------------------------------------
class KrInfo(ctypes.Structure):
    _fields_ = [("n", ctypes.c_int64), ("nnz", ctypes.c_int64), ("outer_iterations", ctypes.c_int32),
                ("inner_iterations", ctypes.c_int32), ("matvecs", ctypes.c_int64), ("boundary_steps", ctypes.c_int64),
                ("residual", ctypes.c_double), ("spmv_seconds", ctypes.c_double), ("spmv_timed", ctypes.c_int64),
                ("value_bytes", ctypes.c_int64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class KrContext:
    """One Knight-Ruiz balancing context on one GPU (fhx_kr_*).  Raises without the built library or without a GPU."""

    def __init__(self, device=0):
        self.L = lib()
        self.h = ctypes.c_void_p()
        rc = self.L.fhx_kr_create(int(device), ctypes.byref(self.h))
        if rc != FHX_OK:
            self.h = None
            raise FhxError(rc, "fhx_kr_create(device=%d) failed: no usable MI355X / HIP runtime" % device)

    def close(self):
        if getattr(self, "h", None):
            self.L.fhx_kr_destroy(self.h)
            self.h = None

    __del__ = close

    def _chk(self, rc):
        if rc != FHX_OK:
            raise FhxError(rc, (self.L.fhx_kr_last_error(self.h) or b"").decode())

    def load_loci(self, chr_ids, mids):
        c, m = _i32(chr_ids), _i32(mids)
        self._chk(self.L.fhx_kr_load_loci(self.h, _ptr(c, ctypes.c_int32), _ptr(m, ctypes.c_int32), len(c)))

    def load_pairs(self, chr1, mid1, chr2, mid2, value):
        a = [_i32(v) for v in (chr1, mid1, chr2, mid2)]
        z = np.ascontiguousarray(value, np.float64)
        bad = ctypes.c_int64(-1)
        rc = self.L.fhx_kr_load_pairs(self.h, *[_ptr(v, ctypes.c_int32) for v in a], _ptr(z, ctypes.c_double), len(z), ctypes.byref(bad))
        if rc == FHX_ERR_REFERENCE_EXIT and bad.value >= 0:
            raise KeyError(bad.value)
        self._chk(rc)

    def shape(self):
        v = [ctypes.c_int64() for _ in range(4)]
        self._chk(self.L.fhx_kr_shape(self.h, *[ctypes.byref(x) for x in v]))
        return tuple(x.value for x in v)

    def get_csr(self, which=0):
        n_full, nnz_full, n_red, nnz_red = self.shape()
        n, nnz = (n_red, nnz_red) if which else (n_full, nnz_full)
        indptr, col, val = np.zeros(n + 1, np.int64), np.zeros(nnz, np.int32), np.zeros(nnz, np.float64)
        self._chk(self.L.fhx_kr_get_csr(self.h, which, _ptr(indptr, ctypes.c_int64), _ptr(col, ctypes.c_int32), _ptr(val, ctypes.c_double)))
        return indptr, col, val

    def row_sums(self):
        out = np.zeros(self.shape()[0], np.float64)
        self._chk(self.L.fhx_kr_row_sums(self.h, _ptr(out, ctypes.c_double)))
        return out

    def remove_sparse(self, perc):
        nrem, val, rem = ctypes.c_int64(), ctypes.c_double(), ctypes.c_int64()
        rc = self.L.fhx_kr_remove_sparse(self.h, float(perc), ctypes.byref(nrem), ctypes.byref(val), ctypes.byref(rem))
        if rc == FHX_ERR_REFERENCE_EXIT:
            raise IndexError((self.L.fhx_kr_last_error(self.h) or b"").decode())
        self._chk(rc)
        idx = np.zeros(nrem.value, np.int64)
        self._chk(self.L.fhx_kr_get_removed(self.h, _ptr(idx, ctypes.c_int64), len(idx), None))
        return idx, val.value, rem.value

    def balance(self, tol=1e-6):
        info = KrInfo()
        rc = self.L.fhx_kr_balance(self.h, float(tol), ctypes.byref(info))
        if rc == FHX_ERR_REFERENCE_EXIT:
            raise ValueError((self.L.fhx_kr_last_error(self.h) or b"").decode())
        self._chk(rc)
        x = np.zeros(info.n, np.float64)
        if info.n:
            self._chk(self.L.fhx_kr_get_x(self.h, _ptr(x, ctypes.c_double)))
        return x, info

    def bias(self):
        out = np.zeros(self.shape()[0], np.float64)
        self._chk(self.L.fhx_kr_bias(self.h, _ptr(out, ctypes.c_double)))
        return out

    def spmv(self, x, which=0, repeats=1):
        x = np.ascontiguousarray(x, np.float64)
        y = np.zeros_like(x)
        sec = ctypes.c_double()
        self._chk(self.L.fhx_kr_spmv(self.h, which, _ptr(x, ctypes.c_double), _ptr(y, ctypes.c_double), int(repeats), ctypes.byref(sec)))
        return y, sec.value

    def dot(self, a, b=None):
        a = np.ascontiguousarray(a, np.float64)
        out = ctypes.c_double()
        bb = np.ascontiguousarray(b, np.float64) if b is not None else None
        self._chk(self.L.fhx_kr_dot(self.h, _ptr(a, ctypes.c_double), _ptr(bb, ctypes.c_double) if bb is not None else None, len(a), ctypes.byref(out)))
        return out.value


# ---- merging of nearby contacts (fhx_cni_*) --------------------------------------------------------------------------
class CniInfo(ctypes.Structure):
    _fields_ = [(k, ctypes.c_int64) for k in ("rows", "nodes", "components", "selected", "pick_rounds", "largest_component")]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


CNI_RECORD = np.dtype([("chr", np.int32), ("reserved", np.int32), ("n_lo", np.int64), ("n_hi", np.int64), ("cc", np.int64),
                       ("p", np.float64), ("q", np.float64), ("box_min_lo", np.int64), ("box_max_lo", np.int64),
                       ("box_min_hi", np.int64), ("box_max_hi", np.int64), ("sum_cc", np.int64), ("box_cells", np.int64),
                       ("component_size", np.int64), ("first_row", np.int64)])


class CniContext:
    """Connected-component merging of significant contacts on one GPU (fhx_cni_*).  Raises without the library or a GPU."""

    def __init__(self, device=0):
        self.L = lib()
        self.h = ctypes.c_void_p()
        rc = self.L.fhx_cni_create(int(device), ctypes.byref(self.h))
        if rc != FHX_OK:
            self.h = None
            raise FhxError(rc, "fhx_cni_create(device=%d) failed: no usable MI355X / HIP runtime" % device)

    def close(self):
        if getattr(self, "h", None):
            self.L.fhx_cni_destroy(self.h)
            self.h = None

    __del__ = close

    def _chk(self, rc):
        if rc != FHX_OK:
            raise FhxError(rc, (self.L.fhx_cni_last_error(self.h) or b"").decode())

    def load(self, chr_ids, n1, n2, cc, p, q, bin_size):
        c = _i32(chr_ids)
        a = [np.ascontiguousarray(v, np.int64) for v in (n1, n2, cc)]
        f = [np.ascontiguousarray(v, np.float64) for v in (p, q)]
        nodes = ctypes.c_int64()
        self._chk(self.L.fhx_cni_load(self.h, _ptr(c, ctypes.c_int32), *[_ptr(v, ctypes.c_int64) for v in a],
                                      *[_ptr(v, ctypes.c_double) for v in f], len(c), int(bin_size), ctypes.byref(nodes)))
        return nodes.value

    def run(self, connectivity=8, top_percent=100, neighborhood=2, sort_order=0):
        info = CniInfo()
        self._chk(self.L.fhx_cni_run(self.h, connectivity, top_percent, neighborhood, sort_order, ctypes.byref(info)))
        out = np.zeros(info.selected, CNI_RECORD)
        if info.selected:
            self._chk(self.L.fhx_cni_get_records(self.h, out.ctypes.data_as(ctypes.c_void_p), len(out), None))
        return out, info
