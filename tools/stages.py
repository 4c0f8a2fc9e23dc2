"""Kernel name -> stage of the step (the names bench.py's STAGE_BYTES and mml_profile use).  One table for every tool that
turns a rocprofv3 CSV into per-stage numbers; longest prefix wins, template arguments and namespaces are stripped first."""
import re

PREFIX = [
    ("k_assign_ends", "assign_ends"), ("k_assign_onepass", "assign_onepass"), ("k_assign_tables", "assign_tables"),
    ("k_seg_records", "assign_tables"),
    ("k_assign_init", "assign_count"), ("k_assign_a", "assign_count"), ("k_assign_b", "assign_scan"),
    ("k_assign_c_direct", "assign_scatter"), ("k_assign_c_staged", "assign_scatter"), ("k_assign_c", "assign_scatter"),
    ("k_stencil_break", "stencil"), ("k_stencil_redo", "stencil"), ("k_stencil", "stencil"), ("k_queue_prefix", "stencil"),
    ("k_select", "select"), ("k_crop", "crop_compact"), ("k_livox_extrinsic", "livox_extrinsic"),
    ("k_undistort_prep", "undistort"), ("k_undistort", "undistort"),
    ("k_voxel", "voxel_downsample"), ("k_seg_", "voxel_downsample"),
    ("k_assoc_prefix", "associate"), ("k_associate_fit_all", "associate_fit"), ("k_associate_fit", "associate_far"),
    ("k_associate_hard", "associate_far"), ("k_associate", "associate"), ("k_assoc_stats", "assoc_stats"),
    ("k_solve", "solve"), ("k_window_round", "solve"), ("k_window_export", "solve"),
]
PREFIX.sort(key=lambda kv: -len(kv[0]))


def short_name(kernel_name, keep_template=False):
    k = kernel_name.replace("(anonymous namespace)::", "").replace("void ", "")
    k = re.sub(r"\(.*", "", k)
    return k if keep_template else k.split("<")[0]


def stage_of(kernel_name):
    k = short_name(kernel_name)
    for p, st in PREFIX:
        if k.startswith(p):
            return st
    return None
