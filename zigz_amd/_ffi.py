"""ctypes binding of include/zigz_hip.h (libzigz_hip.so).  The library is loaded from the in-tree
zigz_amd/lib/ directory; a missing library is an ImportError -- there is no CPU fallback."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libzigz_hip.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `python -m zigz_amd.build` (hipcc, --offload-arch=gfx950). "
        "zigz_amd has no CPU fallback.")

lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)

u64p = C.POINTER(C.c_uint64)
u32p = C.POINTER(C.c_uint32)
u8p = C.POINTER(C.c_uint8)
szp = C.POINTER(C.c_size_t)
vp = C.c_void_p


class LaunchRec(C.Structure):  # zigz_launch_rec
    _fields_ = [("cls", C.c_uint32), ("reserved", C.c_uint32), ("perms", C.c_uint64), ("start_us", C.c_double), ("end_us", C.c_double)]


class KernelStats(C.Structure):
    _fields_ = [("merkle_build_us", C.c_double), ("eval_us", C.c_double), ("path_us", C.c_double),
                ("bind_us", C.c_double), ("bind_launches", C.c_uint64), ("keccak_permutations", C.c_uint64),
                ("bind_vec_us", C.c_double), ("bind_vec_launches", C.c_uint64), ("bind_vec_bytes", C.c_uint64),
                ("run_aware_columns", C.c_uint64), ("run_aware_dense_nodes", C.c_uint64),
                ("run_aware_hashed", C.c_uint64), ("run_aware_us", C.c_double),
                ("cons_columns", C.c_uint64), ("cons_dense_nodes", C.c_uint64), ("cons_hashed", C.c_uint64),
                ("cons_probe_distinct", C.c_uint64),
                ("keccak_leaves_us", C.c_double), ("keccak_leaves_perms", C.c_uint64),
                ("keccak_level_wide_us", C.c_double), ("keccak_level_wide_perms", C.c_uint64),
                ("keccak_level_small_us", C.c_double), ("keccak_level_small_perms", C.c_uint64),
                ("small_domain_columns", C.c_uint64), ("small_domain_us", C.c_double),
                ("small_domain_fallback_waves", C.c_uint64),
                ("structure_us", C.c_double), ("list_hash_us", C.c_double), ("list_hash_perms", C.c_uint64),
                ("top_us", C.c_double), ("top_perms", C.c_uint64), ("rebuilds", C.c_uint64),
                ("eval_constant_columns", C.c_uint64)]


class BenchResult(C.Structure):
    _fields_ = [("avg_us", C.c_double), ("min_us", C.c_double), ("max_us", C.c_double), ("algorithmic_bytes", C.c_uint64),
                ("units", C.c_uint64), ("launches", C.c_uint32)]


ALLGATHER_FN = C.CFUNCTYPE(C.c_int, vp, vp, C.c_size_t, vp)  # zigz_allgather_fn
RADIX_BLOCK_SUMS_FN = C.CFUNCTYPE(C.c_int32, vp, C.c_uint, u64p)
RADIX_FOLD_FN = C.CFUNCTYPE(C.c_int32, vp, C.c_uint, u64p, C.c_uint, u64p)
RADIX_READ_TAIL_FN = C.CFUNCTYPE(C.c_int32, vp, C.c_size_t, u64p)


class RadixOps(C.Structure):  # zigz_radix_ops
    _fields_ = [("user", vp), ("block_sums", RADIX_BLOCK_SUMS_FN), ("fold", RADIX_FOLD_FN), ("read_tail", RADIX_READ_TAIL_FN)]


# name -> (restype, argtypes).  Every symbol include/zigz_hip.h declares must appear here
# (tests/test_abi.py cross-checks this table against the header and the shared object).
SIGNATURES = {
    "zigz_abi_version": (C.c_uint32, []),
    "zigz_status_name": (C.c_char_p, [C.c_int32]),
    "zigz_device_count": (C.c_int32, [C.POINTER(C.c_int)]),
    "zigz_ctx_create": (C.c_int32, [C.c_int, C.POINTER(vp)]),
    "zigz_ctx_destroy": (None, [vp]),
    "zigz_last_error": (C.c_char_p, [vp]),
    "zigz_ctx_set_stream": (C.c_int32, [vp, vp]),
    "zigz_ctx_get_stream": (vp, [vp]),
    "zigz_ctx_synchronize": (C.c_int32, [vp]),
    "zigz_dev_alloc": (C.c_int32, [vp, C.c_size_t, C.POINTER(vp)]),
    "zigz_dev_free": (C.c_int32, [vp, vp]),
    "zigz_dev_upload_u64": (C.c_int32, [vp, u64p, C.c_size_t, vp]),
    "zigz_dev_download_u64": (C.c_int32, [vp, vp, C.c_size_t, u64p]),
    "zigz_mle_bind": (C.c_int32, [vp, u64p, C.c_size_t, C.c_uint64, u64p]),
    "zigz_mle_round_poly": (C.c_int32, [vp, u64p, C.c_size_t, u64p]),
    "zigz_mle_sum": (C.c_int32, [vp, u64p, C.c_size_t, u64p]),
    "zigz_mle_eval": (C.c_int32, [vp, u64p, C.c_size_t, u64p, C.c_size_t, u64p]),
    "zigz_sumcheck_prove": (C.c_int32, [vp, u64p, C.c_size_t, u64p, u64p, u64p]),
    "zigz_sumcheck_prove_interactive": (C.c_int32, [vp, u64p, C.c_size_t, u64p, C.c_size_t, u64p, u64p, u64p]),
    "zigz_merkle_commit": (C.c_int32, [vp, u64p, C.c_size_t, u8p, szp, C.POINTER(vp)]),
    "zigz_merkle_open": (C.c_int32, [vp, vp, C.c_size_t, u8p, u8p, u64p]),
    "zigz_merkle_destroy": (None, [vp, vp]),
    "zigz_commit_open": (C.c_int32, [vp, u64p, C.c_size_t, vp, u64p, C.c_size_t, u64p, u64p, u8p, u8p, u64p]),
    "zigz_commit_begin": (C.c_int32, [vp, u64p, C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(vp)]),
    "zigz_commit_begin_dev": (C.c_int32, [vp, vp, C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(vp)]),
    "zigz_commit_begin_batch": (C.c_int32, [vp, C.POINTER(vp), C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(vp)]),
    "zigz_commit_roots": (C.c_int32, [vp, u8p]),
    "zigz_commit_open_all": (C.c_int32, [vp, u64p, u64p, u64p, u64p, u8p, u8p]),
    "zigz_commit_job_tree": (C.c_int32, [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]),
    "zigz_commit_end": (None, [vp]),
    "zigz_lasso_prove": (C.c_int32, [vp, u64p, C.c_size_t, u64p, C.c_size_t, C.c_size_t, C.c_size_t, szp, u64p, u64p,
                                     u64p, u8p, u8p]),
    "zigz_lasso_prove_with_mapping": (C.c_int32, [vp, u64p, C.c_size_t, u64p, C.c_size_t, C.c_size_t, C.c_size_t, u64p,
                                                  C.c_size_t, szp, u64p, u64p, u64p, u8p, u8p]),
    "zigz_lasso_fingerprints": (C.c_int32, [vp, u64p, C.c_size_t, C.c_size_t, u64p]),
    "zigz_dev_reduce_u64": (C.c_int32, [vp, u64p, C.c_size_t, vp]),
    "zigz_dev_witness_from_rows": (C.c_int32, [vp, u64p, C.c_size_t, C.c_size_t, vp, C.c_size_t]),
    "zigz_dev_witness_from_steps": (C.c_int32, [vp, vp, C.c_size_t, C.c_size_t, u64p, vp, C.c_size_t]),
    "zigz_dev_witness_from_steps_async": (C.c_int32, [vp, vp, C.c_size_t, C.c_size_t, u64p, vp, C.c_size_t]),
    "zigz_dev_witness_from_steps32": (C.c_int32, [vp, vp, C.c_size_t, vp, C.c_size_t, C.c_size_t, u64p, vp, C.c_size_t]),
    "zigz_dev_witness_from_steps32_ws": (C.c_int32, [vp, vp, C.c_size_t, vp, C.c_size_t, C.c_size_t, u64p, C.POINTER(vp), szp]),
    "zigz_dev_witness_from_steps16": (C.c_int32, [vp, vp, C.c_size_t, vp, C.c_size_t, C.c_uint64, vp, C.c_size_t, C.c_size_t, u64p, vp, C.c_size_t]),
    "zigz_dev_witness_from_steps16_ws": (C.c_int32, [vp, vp, C.c_size_t, vp, C.c_size_t, C.c_uint64, vp, C.c_size_t, C.c_size_t, u64p,
                                                     C.POINTER(vp), szp]),
    "zigz_dev_witness_from_steps_ws": (C.c_int32, [vp, vp, C.c_size_t, C.c_size_t, u64p, C.POINTER(vp), szp]),
    "zigz_dev_mem_info": (C.c_int32, [vp, szp, szp]),
    "zigz_ctx_set_epoch": (C.c_int32, [vp, vp]),
    "zigz_ctx_launch_log": (C.c_int32, [vp, C.POINTER(LaunchRec), C.c_size_t, szp]),
    "zigz_ctx_release_workspaces": (C.c_int32, [vp]),
    "zigz_host_register": (C.c_int32, [vp, vp, C.c_size_t]),
    "zigz_host_unregister": (C.c_int32, [vp, vp]),
    "zigz_dev_mle_bind": (C.c_int32, [vp, vp, C.c_size_t, C.c_uint64, vp]),
    "zigz_dev_mle_bind_sums": (C.c_int32, [vp, vp, C.c_size_t, C.c_uint64, vp, u64p]),
    "zigz_dev_mle_half_sums": (C.c_int32, [vp, vp, C.c_size_t, u64p]),
    "zigz_dev_mle_eval": (C.c_int32, [vp, vp, C.c_size_t, u64p, C.c_size_t, u64p]),
    "zigz_dev_sumcheck_prove": (C.c_int32, [vp, vp, C.c_size_t, vp, u64p, u64p, u64p, u64p]),
    "zigz_dev_sumcheck_prove_sharded": (C.c_int32, [vp, vp, C.c_size_t, C.c_int, C.c_int, ALLGATHER_FN, vp, u64p, u64p, u64p]),
    "zigz_sumcheck_radix_run": (C.c_int32, [C.POINTER(RadixOps), C.c_size_t, C.c_int, C.c_int, ALLGATHER_FN, vp, u64p, u64p,
                                            u64p, u64p]),
    "zigz_sumcheck_radix_run_reduced": (C.c_int32, [C.POINTER(RadixOps), C.c_size_t, C.c_int, C.c_int, ALLGATHER_FN, vp, u64p, u64p,
                                                    u64p, u64p]),
    "zigz_shm_comm_create": (C.c_int32, [C.c_char_p, C.c_int, C.c_int, C.c_size_t, C.c_double, C.POINTER(vp)]),
    "zigz_shm_allgather": (C.c_int, [vp, vp, C.c_size_t, vp]),
    "zigz_shm_comm_destroy": (None, [vp]),
    "zigz_rccl_unique_id": (C.c_int32, [u8p]),
    "zigz_rccl_comm_create": (C.c_int32, [C.c_int, u8p, C.c_int, C.c_int, C.c_size_t, C.POINTER(vp)]),
    "zigz_rccl_allgather": (C.c_int, [vp, vp, C.c_size_t, vp]),
    "zigz_rccl_allreduce_u64": (C.c_int, [vp, u64p, C.c_size_t, u64p]),
    "zigz_rccl_allreduce_u64_dev": (C.c_int, [vp, vp, C.c_size_t, vp]),
    "zigz_rccl_stream_wait": (C.c_int, [vp, vp]),
    "zigz_rccl_comm_set_timeout": (None, [vp, C.c_double]),
    "zigz_rccl_comm_abort": (None, [vp]),
    "zigz_rccl_comm_rank": (C.c_int, [vp]),
    "zigz_rccl_comm_world": (C.c_int, [vp]),
    "zigz_rccl_comm_destroy": (None, [vp]),
    "zigz_dev_sumcheck_prove_rccl": (C.c_int32, [vp, vp, C.c_size_t, vp, u64p, u64p, u64p]),
    "zigz_transcript_new": (vp, []),
    "zigz_transcript_free": (None, [vp]),
    "zigz_transcript_append_bytes": (None, [vp, C.c_char_p, C.c_size_t]),
    "zigz_transcript_append_field": (None, [vp, C.c_uint64]),
    "zigz_transcript_append_tagged_counter": (None, [vp, C.c_char_p, C.c_size_t, C.c_uint64, C.c_uint64]),
    "zigz_transcript_challenge": (C.c_uint64, [vp]),
    "zigz_sha3_256": (None, [C.c_char_p, C.c_size_t, u8p]),
    "zigz_sha256": (None, [C.c_char_p, C.c_size_t, u8p]),
    "zigz_host_keccak_impl": (C.c_char_p, []),
    "zigz_device_set_blocking_sync": (C.c_int32, [C.c_int, C.c_int]),
    "zigz_host_sponge_servers": (None, [C.c_int]),
    "zigz_host_sponge_batching": (C.c_int, []),
    "zigz_host_keccak_permute_x8": (None, [u64p]),
    "zigz_host_keccak_permute": (None, [u64p, C.c_int]),
    "zigz_ctx_set_option": (C.c_int32, [vp, C.c_char_p, C.c_int64]),
    "zigz_ctx_get_option": (C.c_int32, [vp, C.c_char_p, C.POINTER(C.c_int64)]),
    "zigz_ctx_enable_timing": (C.c_int32, [vp, C.c_int]),
    "zigz_bench_kernel": (C.c_int32, [vp, C.c_char_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.POINTER(BenchResult)]),
    "zigz_ctx_get_stats": (C.c_int32, [vp, C.POINTER(KernelStats)]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _f = getattr(lib, _name)  # AttributeError here = the .so does not export a declared symbol
    _f.restype = _res
    _f.argtypes = _args
