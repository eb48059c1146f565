"""MI355X-native batched Scotland-Yard environment engine (env.step hot path of
elte-collective-intelligence/student-mechanism-design) behind the reference's own interfaces.

Product path = HIP kernels in csrc/ reached through the C ABI (include/sy_env.h).  Nothing here
falls back to CPU: without libsy_env.so (build with `python -m student_mechanism_design_amd.build`)
device classes raise `EngineError`.
"""
from ._lib import EngineError, LIB_PATH  # noqa: F401
from .graph import (Board, PackedPool, make_board, sample_board, sample_board_pool, pack_pool,  # noqa: F401
                    pack_ell, all_pairs_shortest_paths, device_all_pairs_shortest_paths, reward_tables,
                    node_stride_for, DevicePool, sample_board_pool_device)
from .env import BatchedScotlandYardEnv, REWARD_WEIGHT_NAMES, DEFAULT_ACTION, weights_to_array  # noqa: F401
from .action_mask import compute_action_mask, get_action_mask_for_agent, ActionMaskResult  # noqa: F401
from .belief import DeviceBeliefTracker  # noqa: F401
