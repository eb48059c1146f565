"""ctypes binding of the C ABI in include/sy_env.h.  There is NO fallback path: if the HIP library
is missing or a call fails, this module raises."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SY_ENGINE_LIB") or os.path.join(HERE, "libsy_env.so")  # SY_ENGINE_LIB: diagnostic builds

ELL_WIDTH = 16
MAX_AGENTS = 8
MAX_NODES = 1024
NUM_WEIGHTS = 11
MRX_MONEY = 1000
ABI_VERSION = 7
STATUS_BELIEF_WAIT_EXPIRED = 1
STATUS_RING_WAIT_EXPIRED = 2


class EngineError(RuntimeError):
    pass


class EnvConfig(C.Structure):
    _fields_ = [("num_envs", C.c_int32), ("num_nodes", C.c_int32), ("num_police", C.c_int32),
                ("agent_money", C.c_int32), ("max_timestep", C.c_int32), ("num_graphs", C.c_int32),
                ("node_stride", C.c_int32), ("reveal_interval", C.c_int32), ("police_evidence", C.c_int32),
                ("belief_init_onehot", C.c_int32), ("auto_reset", C.c_int32), ("waves_per_block", C.c_int32),
                ("env_id_offset", C.c_uint64)]


class EnvState(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("pos", "budget", "t", "step_count", "visits", "belief", "mask",
                                          "reward", "terminated", "truncated", "winner")]


class MappoWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("w1t", "b1", "w2t", "b2", "c1t", "cb1", "c2", "cb2", "w2", "logit_bound")]


class ReturnsArgs(C.Structure):
    _fields_ = [("T", C.c_int32), ("B", C.c_int32), ("A", C.c_int32), ("mode", C.c_int32), ("reward", C.c_void_p),
                ("reward_f64", C.c_int32), ("reward_stride_t", C.c_int64), ("reward_stride_b", C.c_int64),
                ("done_a", C.c_void_p), ("done_b", C.c_void_p), ("done_bytes", C.c_int32),
                ("done_stride_t", C.c_int64), ("done_stride_b", C.c_int64), ("value", C.c_void_p),
                ("value_stride_t", C.c_int64), ("value_stride_b", C.c_int64), ("value_stride_a", C.c_int64),
                ("last_value", C.c_void_p), ("last_value_stride_b", C.c_int64), ("last_value_stride_a", C.c_int64),
                ("gamma", C.c_double), ("lam", C.c_double), ("compute_f64", C.c_int32), ("returns", C.c_void_p),
                ("adv", C.c_void_p)]


class PpoPackArgs(C.Structure):
    _fields_ = [("record", C.c_void_p), ("record_words", C.c_int32), ("log_prob", C.c_void_p), ("adv", C.c_void_p),
                ("team_ret", C.c_void_p), ("rows", C.c_void_p), ("row0", C.c_int32), ("num_rows", C.c_int64),
                ("num_envs", C.c_int32), ("env_graph", C.c_void_p), ("num_police", C.c_int32), ("image", C.c_void_p),
                ("image_bytes", C.c_int64), ("shuffle_domain", C.c_int64), ("shuffle_seed", C.c_uint64), ("chunk_rows", C.c_int64),
                ("record_chunk_stride", C.c_int64), ("log_prob_chunk_stride", C.c_int64)]


class PpoArgs(C.Structure):
    _fields_ = [("image", C.c_void_p), ("image_rows", C.c_int64), ("row0", C.c_int32), ("row0_dev", C.c_void_p),
                ("num_rows", C.c_int32), ("ell", C.c_void_p), ("num_police", C.c_int32), ("num_nodes", C.c_int32),
                ("hidden", C.c_int32), ("params", C.c_void_p), ("clip", C.c_float), ("value_coef", C.c_float),
                ("scratch", C.c_void_p), ("scratch_floats", C.c_int64), ("grads", C.c_void_p), ("adam_m", C.c_void_p),
                ("adam_v", C.c_void_p), ("adam_step", C.c_void_p), ("lr", C.c_float), ("beta1", C.c_float),
                ("beta2", C.c_float), ("eps", C.c_float)]


class RolloutBuffers(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("record", "mask", "belief", "log_prob")]


EXPORTS = ["sy_abi_version", "sy_record_words", "sy_last_error", "sy_env_create", "sy_env_destroy", "sy_env_launch_info",
           "sy_env_set_graph_pool", "sy_env_set_rewards", "sy_env_set_policy", "sy_env_bind_state", "sy_env_reset", "sy_env_reset_to",
           "sy_env_step", "sy_env_step_record", "sy_env_rollout", "sy_action_mask_dense", "sy_belief_update", "sy_build_apsp", "sy_sample_boards",
           "sy_masked_categorical_sample", "sy_mappo_policy_act", "sy_env_bind_status", "sy_env_status",
           "sy_returns_advantages", "sy_build_id", "sy_env_rollout_kernel_name",
           "sy_gnn_padded_features", "sy_gnn_param_floats", "sy_gnn_q_act",
           "sy_ppo_slab_floats", "sy_ppo_scratch_floats", "sy_mappo_ppo_grad", "sy_ppo_image_bytes", "sy_ppo_pack", "sy_ppo_adam_step", "sy_env_set_belief_layout"]

_lib = None


def load():
    """Load libsy_env.so (built by student_mechanism_design_amd.build).  Raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EngineError(
            f"HIP engine library not found at {LIB_PATH}. Build it with "
            "`python -m student_mechanism_design_amd.build` (needs hipcc); there is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    vp, i32, u64 = C.c_void_p, C.c_int32, C.c_uint64
    lib.sy_abi_version.restype = C.c_int
    lib.sy_last_error.restype = C.c_char_p
    lib.sy_build_id.restype = C.c_char_p
    lib.sy_gnn_padded_features.argtypes = [i32]
    lib.sy_gnn_param_floats.argtypes = [i32]
    lib.sy_gnn_q_act.argtypes = [vp, vp, C.c_int64, vp, C.c_int64, vp, i32, vp, vp, vp, i32, i32, i32, i32, C.c_float, u64, u64,
                                 vp, vp, vp, vp]
    lib.sy_env_rollout_kernel_name.argtypes = [vp, i32, C.c_char_p, i32]
    lib.sy_record_words.argtypes = [i32]
    lib.sy_env_create.argtypes = [C.POINTER(EnvConfig), C.POINTER(vp)]
    lib.sy_env_destroy.argtypes = [vp]
    lib.sy_env_launch_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    lib.sy_env_set_graph_pool.argtypes = [vp, vp, vp, vp, vp, i32]
    lib.sy_env_set_belief_layout.argtypes = [vp, vp, vp]
    lib.sy_env_set_rewards.argtypes = [vp, C.POINTER(C.c_double), vp, i32, vp, i32]
    lib.sy_env_bind_state.argtypes = [vp, C.POINTER(EnvState)]
    lib.sy_env_set_policy.argtypes = [vp, C.POINTER(MappoWeights), i32]
    lib.sy_env_bind_status.argtypes = [vp, vp]
    lib.sy_env_status.argtypes = [vp, vp, C.POINTER(C.c_uint32)]
    lib.sy_returns_advantages.argtypes = [C.POINTER(ReturnsArgs), vp]
    lib.sy_env_reset.argtypes = [vp, vp, u64, vp]
    lib.sy_env_reset_to.argtypes = [vp, vp, vp]
    lib.sy_env_step.argtypes = [vp, vp, vp]
    lib.sy_env_step_record.argtypes = [vp, vp, C.POINTER(RolloutBuffers), vp]
    lib.sy_env_rollout.argtypes = [vp, i32, C.POINTER(RolloutBuffers), vp]
    lib.sy_action_mask_dense.argtypes = [vp, vp, vp, i32, vp, vp, i32, vp, vp]
    lib.sy_belief_update.argtypes = [vp, vp, i32, i32, vp, vp, i32, vp, i32, vp]
    lib.sy_build_apsp.argtypes = [vp, i32, i32, vp, vp]
    lib.sy_sample_boards.argtypes = [i32, i32, i32, i32, u64, i32, vp, vp, vp, vp, vp, i32, vp]
    lib.sy_masked_categorical_sample.argtypes = [vp, C.c_int64, vp, C.c_int64, i32, i32, u64, u64, vp, i32, vp, vp, vp, vp]
    lib.sy_mappo_policy_act.argtypes = [vp, vp, C.c_int64, C.POINTER(MappoWeights), i32, i32, i32, i32, u64, u64, vp, vp, vp, vp, vp, vp]
    lib.sy_ppo_slab_floats.argtypes = [i32, i32]
    lib.sy_ppo_scratch_floats.argtypes = [i32, i32, i32]
    lib.sy_mappo_ppo_grad.argtypes = [C.POINTER(PpoArgs), vp]
    lib.sy_ppo_image_bytes.argtypes = [i32, C.c_int64]
    lib.sy_ppo_pack.argtypes = [C.POINTER(PpoPackArgs), vp]
    lib.sy_ppo_adam_step.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, C.c_float, C.c_float, C.c_float, C.c_float, vp]
    for name in EXPORTS:
        fn = getattr(lib, name)
        if name not in ("sy_last_error", "sy_build_id"):
            fn.restype = C.c_int
    lib.sy_ppo_scratch_floats.restype = C.c_int64
    lib.sy_ppo_image_bytes.restype = C.c_int64
    if lib.sy_abi_version() != ABI_VERSION:
        raise EngineError(f"libsy_env.so ABI {lib.sy_abi_version()} != expected {ABI_VERSION}; rebuild it")
    _lib = lib
    return lib


def build_id() -> str:
    """Digest of the sources the loaded library was built from (sy_build_id)."""
    return load().sy_build_id().decode()


def check(rc, what=""):
    if rc != 0:
        msg = load().sy_last_error().decode("utf-8", "replace")
        raise EngineError(f"{what} failed ({rc}): {msg}")
