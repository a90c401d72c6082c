"""ctypes binding of include/katome_gpu.h (the same symbols a Rust `extern "C"` block binds)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)

STATUS = {0: "OK", -1: "E_PATH", -2: "E_IS_DIR", -3: "E_NOT_EXIST", -4: "E_OPEN", -5: "E_PARSE", -6: "E_SHORT_READ",
          -7: "E_ARG", -8: "E_DEVICE", -9: "E_OOM", -10: "E_UNSUPPORTED"}


class Settings(C.Structure):
    _fields_ = [("k", C.c_uint32), ("file_type", C.c_uint8), ("reverse_complement", C.c_uint8), ("flags", C.c_uint16),
                ("min_weight", C.c_uint32), ("device", C.c_int32), ("table_slots_hint", C.c_uint64),
                ("n_devices", C.c_int32), ("_reserved", C.c_uint32)]


class Graph(C.Structure):
    _fields_ = [("n_nodes", C.c_uint64), ("n_edges", C.c_uint64), ("read_bytes", C.c_uint64), ("k", C.c_uint32),
                ("key_words", C.c_uint32), ("label_stride", C.c_uint32), ("_pad", C.c_uint32),
                ("edge_src", u64p), ("edge_dst", u64p), ("edge_weight", u32p), ("edge_label", u8p),
                ("edge_key", u64p), ("node_key", u64p), ("edge_age", u32p)]


class Stats(C.Structure):
    _fields_ = [("node_count", C.c_uint64), ("edge_count", C.c_uint64), ("max_edge_weight", C.c_uint32),
                ("_pad", C.c_uint32), ("avg_edge_weight", C.c_double), ("max_in_degree", C.c_uint64),
                ("max_out_degree", C.c_uint64), ("avg_out_degree", C.c_double), ("incoming_vert_count", C.c_uint64),
                ("outgoing_vert_count", C.c_uint64)]


class Reads(C.Structure):
    _fields_ = [("n_records", C.c_uint64), ("n_reads", C.c_uint64), ("read_bytes", C.c_uint64),
                ("packed_bytes", C.c_uint64), ("total_windows", C.c_uint64), ("fixed_len", C.c_uint32),
                ("_pad", C.c_uint32), ("packed", u8p), ("byte_off", u64p), ("len", u32p)]


class PruneStats(C.Structure):
    _fields_ = [("passes", C.c_uint64), ("walks", C.c_uint64), ("dead_walks", C.c_uint64), ("marked", C.c_uint64),
                ("removed_edges", C.c_uint64), ("removed_by_duplicates", C.c_uint64), ("removed_nodes", C.c_uint64),
                ("host_ms", C.c_double), ("total_ms", C.c_double)]


class Contigs(C.Structure):
    _fields_ = [("n_nodes", C.c_uint64), ("n_edges", C.c_uint64), ("label_bytes", C.c_uint64), ("read_bytes", C.c_uint64),
                ("k", C.c_uint32), ("key_words", C.c_uint32), ("edge_src", u64p), ("edge_dst", u64p), ("edge_weight", u32p),
                ("edge_kmers", u32p), ("edge_label_off", u64p), ("edge_label", u8p), ("node_key", u64p)]


class DevContigs(C.Structure):
    _fields_ = [("n_nodes", C.c_uint64), ("n_edges", C.c_uint64), ("label_bytes", C.c_uint64), ("key_words", C.c_uint32),
                ("_pad", C.c_uint32), ("d_edge_src", C.c_void_p), ("d_edge_dst", C.c_void_p), ("d_edge_weight", C.c_void_p),
                ("d_edge_kmers", C.c_void_p), ("d_edge_label_off", C.c_void_p), ("d_edge_label", C.c_void_p),
                ("d_node_key", C.c_void_p)]


class DevGraph(C.Structure):
    _fields_ = [("n_nodes", C.c_uint64), ("n_edges", C.c_uint64), ("key_words", C.c_uint32),
                ("label_stride", C.c_uint32), ("d_edge_key", C.c_void_p), ("d_edge_weight", C.c_void_p),
                ("d_edge_src", C.c_void_p), ("d_edge_dst", C.c_void_p), ("d_edge_label", C.c_void_p),
                ("d_node_key", C.c_void_p), ("d_edge_age", C.c_void_p)]


class DistGraph(C.Structure):
    _fields_ = [("n_edges", C.c_uint64), ("n_nodes", C.c_uint64), ("total_edges", C.c_uint64), ("total_nodes", C.c_uint64),
                ("node_base", C.c_uint64), ("key_words", C.c_uint32), ("label_stride", C.c_uint32),
                ("d_edge_key", C.c_void_p), ("d_edge_weight", C.c_void_p), ("d_edge_src", C.c_void_p),
                ("d_edge_dst", C.c_void_p), ("d_edge_label", C.c_void_p), ("d_node_key", C.c_void_p),
                ("d_edge_id", C.c_void_p), ("d_node_id", C.c_void_p), ("d_edge_age", C.c_void_p)]


# the caller's transport (katome_comm_callbacks)
A2A_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, u64p, u64p, C.c_void_p, u64p, u64p, C.c_uint64, C.c_int)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, u64p, C.c_uint64, C.c_int)


class CommCallbacks(C.Structure):
    _fields_ = [("user", C.c_void_p), ("alltoallv", A2A_FN), ("allreduce_u64", ALLREDUCE_FN)]


# every symbol include/katome_gpu.h declares: name -> (restype, argtypes)
_vp, _sz, _i, _u32, _u64, _dbl = C.c_void_p, C.c_size_t, C.c_int, C.c_uint32, C.c_uint64, C.c_double
_pp = C.POINTER(C.c_char_p)
SYMBOLS = {
    "katome_build_files": (_i, [C.POINTER(Settings), _pp, _sz, C.POINTER(C.POINTER(Graph))]),
    "katome_build_files_staged": (_i, [C.POINTER(Settings), _pp, _sz, C.c_char_p, _u64, C.POINTER(C.POINTER(Graph))]),
    "katome_build_packed": (_i, [C.POINTER(Settings), _vp, _u64, _u32, _vp, C.POINTER(C.POINTER(Graph))]),
    "katome_graph_free": (None, [C.POINTER(Graph)]),
    "katome_graph_stats": (_i, [C.POINTER(Graph), C.POINTER(Stats)]),
    "katome_last_error": (C.c_char_p, []),
    "katome_abi_version": (_u32, []),
    "katome_ingest_files": (_i, [C.POINTER(Settings), _pp, _sz, C.POINTER(C.POINTER(Reads))]),
    "katome_reads_free": (None, [C.POINTER(Reads)]),
    "katome_builder_create": (_i, [C.POINTER(Settings), C.POINTER(_vp)]),
    "katome_builder_destroy": (None, [_vp]),
    "katome_record_words": (_u32, [_u32]),
    "katome_builder_counts": (_i, [_vp, u64p]),
    "katome_builder_profile": (_i, [_vp, _i]),
    "katome_builder_profile_read": (_i, [_vp, C.POINTER(_dbl), u64p]),
    "katome_builder_profile_read_work": (_i, [_vp, C.POINTER(_dbl), u64p, u64p]),
    "katome_phase_count": (_u32, []),
    "katome_phase_name": (C.c_char_p, [_u32]),
    "katome_dev_extract_fixed": (_i, [_vp, _vp, _u64, _u32, _vp, _vp, _vp]),
    "katome_dev_extract_var": (_i, [_vp, _vp, _u64, _vp, _vp, _vp, _u64, _u64, _vp, _vp]),
    "katome_dev_extract_var_tiles": (_i, [_vp, _vp, _u64, _vp, _vp, _vp, _vp, _u64, _u64, _u64, _u32, _vp, _vp]),
    "katome_dev_extract_var_remainder": (_i, [_vp, _vp, _u64, _vp, _vp, _vp, _vp, _u64, _u64, _u64, _u32, _vp, _vp]),
    "katome_dev_partition": (_i, [_i, _vp, _vp, _u64, _u32, _u32, _vp, _vp, u64p, _vp]),
    "katome_dev_partition_core": (_i, [_i, _vp, _vp, _u64, _u32, _u32, _u32, _u32, _vp, _vp, u64p, _vp]),
    "katome_key_owner": (_u32, [u64p, _u32, _u32, _u32, _u32]),
    "katome_dev_source_ids": (_i, [_i, _vp, _u64, _u32, _vp, _vp, u64p, _vp]),
    "katome_dev_insert": (_i, [_vp, _vp, _u64, _vp]),
    "katome_dev_insert_weighted": (_i, [_vp, _vp, _vp, _u64, _vp]),
    "katome_dev_remove_weak_edges": (_i, [_vp, _u32, _vp]),
    "katome_dev_table_count": (_i, [_vp, u64p, _vp]),
    "katome_tile_span": (_u32, [_u32, _u32]),
    "katome_tile_words": (_u32, [_u32, _u32]),
    "katome_tile_plan": (_u32, [_u32, _u32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "katome_tile_plan_limited": (_u32, [_u32, _u32, _u32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "katome_dev_extract_remainder": (_i, [_vp, _vp, _u64, _u32, _u32, _vp, _vp, _vp]),
    "katome_dev_extract_tiles": (_i, [_vp, _vp, _u64, _u32, _u32, _vp, _vp, _vp]),
    "katome_dev_insert_tiles": (_i, [_vp, _vp, _u64, _u32, _vp]),
    "katome_dev_count_tiles": (_i, [_vp, _vp, _u64, _u32, _u32, _vp, _vp]),
    "katome_dev_expand_tiles": (_i, [_vp, C.POINTER(_vp), C.POINTER(_vp), u64p, _vp]),
    "katome_dev_finalize": (_i, [_vp, C.POINTER(DevGraph), _vp]),
    "katome_shrink_files": (_i, [C.POINTER(Settings), _pp, _sz, C.POINTER(C.POINTER(Contigs))]),
    "katome_shrink_packed": (_i, [C.POINTER(Settings), _vp, _u64, _u32, _vp, C.POINTER(C.POINTER(Contigs))]),
    "katome_contigs_free": (None, [C.POINTER(Contigs)]),
    "katome_dev_standardize_contigs": (_i, [_vp, _vp]),
    "katome_dev_standardize_edges": (_i, [_vp, _u64, _u32, _vp]),
    "katome_dev_shrink": (_i, [_vp, C.POINTER(DevContigs), _vp]),
    "katome_dev_shrink_mode": (_i, [_vp, C.c_uint32, C.POINTER(DevContigs), C.POINTER(C.c_double), _vp]),
    "katome_dev_current_graph": (_i, [_vp, C.POINTER(DevGraph)]),
    "katome_dev_scan_counts": (_i, [_i, _vp, _u64, _vp, _vp]),
    "katome_dev_replay_node_removals": (_i, [_i, _vp, _u64, _u64, _vp, _vp, _vp, _vp]),
    "katome_dev_replay_edge_removals": (_i, [_i, _vp, _vp, _u64, _u64, _vp, _vp, _vp, _vp, _vp]),
    "katome_dev_replay_node_removals64": (_i, [_i, _vp, _u64, _u64, _vp, _vp, _vp, _vp]),
    "katome_dev_replay_edge_removals64": (_i, [_i, _vp, _vp, _u64, _u64, _vp, _vp, _vp, _vp, _vp]),
    "katome_dev_remove_dead_paths": (_i, [_vp, C.POINTER(DevGraph), C.POINTER(PruneStats), _vp]),
    "katome_dev_edges": (_i, [_vp, C.POINTER(_vp), C.POINTER(_vp), u64p, _vp]),
    "katome_dev_release_cache": (_i, [_i]),
    "katome_dev_cache_stats": (_i, [_i, u64p]),
    "katome_dev_sort": (_i, [_i, _vp, _vp, _u64, _u32, _u32, _vp]),
    "katome_dev_unique": (_i, [_i, _vp, _u64, _u32, u64p, _vp]),
    "katome_dev_rank": (_i, [_i, _vp, _u64, _u32, _u32, _vp, _u64, _vp, _vp]),
    "katome_dev_node_ids": (_i, [_i, _vp, _u64, _u32, _vp, _vp, _vp, u64p, _vp]),
    "katome_dev_endpoints": (_i, [_i, _vp, _u64, _u32, _vp, _vp, _vp]),
    "katome_dev_labels": (_i, [_i, _vp, _u64, _u32, _vp, _vp]),
    "katome_dev_synth_reads": (_i, [_i, _u64, _u64, _u32, _u64, _dbl, _u32, _vp, _vp, _vp]),
    # multi-GPU: communicators and the sharded build
    "katome_comm_unique_id": (_i, [_vp]),
    "katome_comm_create_rccl": (_i, [_vp, _i, _i, _i, C.POINTER(_vp)]),
    "katome_comm_create_callbacks": (_i, [C.POINTER(CommCallbacks), _i, _i, _i, C.POINTER(_vp)]),
    "katome_comm_destroy": (None, [_vp]),
    "katome_comm_rank": (_i, [_vp]),
    "katome_comm_world": (_i, [_vp]),
    "katome_comm_kind": (C.c_char_p, [_vp]),
    "katome_comm_set_max_message_bytes": (_i, [_vp, _u64]),
    "katome_comm_allreduce_u64": (_i, [_vp, u64p, _u64, _i]),
    "katome_comm_exchange": (_i, [_vp, _vp, u64p, _vp, _u64, u64p, _u64, _i, _vp]),
    "katome_dist_create": (_i, [C.POINTER(Settings), _vp, C.POINTER(_vp)]),
    "katome_dist_destroy": (None, [_vp]),
    "katome_dist_add_reads": (_i, [_vp, _vp, _u64, _u64, _u32, _vp, _u64, _vp]),
    "katome_dist_remove_weak_edges": (_i, [_vp, _u32]),
    "katome_dist_finalize": (_i, [_vp, C.POINTER(DistGraph), _vp]),
    "katome_dist_gather": (_i, [_vp, _i, C.POINTER(_vp), _vp]),
    "katome_dist_inner": (_vp, [_vp]),
    "katome_dist_route": (C.c_char_p, [_vp]),
    "katome_dist_remove_dead_paths": (_i, [_vp, C.POINTER(DistGraph), C.POINTER(PruneStats), _vp]),
    "katome_dist_current_graph": (_i, [_vp, C.POINTER(DistGraph)]),
    "katome_dist_exchange_count": (_u32, []),
    "katome_dist_exchange_name": (C.c_char_p, [_u32]),
    "katome_dist_exchange_read": (_i, [_vp, u64p]),
    "katome_shard_range": (None, [_u64, _u32, _u32, u64p, u64p]),
}


def lib_path():
    return os.environ.get("KATOME_LIB") or os.path.join(_HERE, "lib", "libkatome_gpu.so")


def lib():
    """Load the HIP library.  Fails loudly when it has not been built: there is no fallback."""
    global _LIB
    if _LIB is None:
        path = lib_path()
        if not os.path.exists(path):
            raise ImportError("%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "or `make -C katome_amd/csrc` (no CPU fallback exists)" % path)
        L = C.CDLL(path)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def last_error():
    return lib().katome_last_error().decode(errors="replace")
