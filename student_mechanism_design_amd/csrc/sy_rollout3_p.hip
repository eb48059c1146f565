// sy_rollout3_p.hip — rollout3 instances with the learned policy in the move wave (sy_env_set_policy), boards of 129..256 nodes: the half-wave scan for 2, 4, 5, 6, 7 police (columns per lane to cover rows of up to 16 neighbours), the paired scan for other counts
// (add an instance here AND in sy_dispatch.hip::plan_rollout)
#include "sy_rollout3.hpp"

namespace sy {

template <int NR, bool REC, int PT, bool POL, int HS>
static bool try_launch(const RolloutPlan& pl, const EngineParams& p, int T, const sy_rollout_buffers& out, int blocks, hipStream_t stream) {
    if (pl.nr != NR || pl.rec != REC || pl.pt != PT || pl.pol != POL || pl.hs != HS) return false;
    hipLaunchKernelGGL((rollout3_kernel<NR, REC, PT, POL, HS>), dim3(blocks), dim3(pl.threads), pl.lds, stream, p, T, out);
    return true;
}

bool launch_r3_p(const RolloutPlan& pl, const EngineParams& p, int T, const sy_rollout_buffers& out, int blocks, hipStream_t stream) {
    return try_launch<4, true, 2, true, 2>(pl, p, T, out, blocks, stream) ||
           try_launch<4, true, 4, true, 2>(pl, p, T, out, blocks, stream) ||
           try_launch<4, true, 4, true, 3>(pl, p, T, out, blocks, stream) ||
           try_launch<4, true, 5, true, 2>(pl, p, T, out, blocks, stream) ||
           try_launch<4, true, 5, true, 4>(pl, p, T, out, blocks, stream) ||
           try_launch<4, true, 6, true, 3>(pl, p, T, out, blocks, stream) ||
           try_launch<4, true, 6, true, 4>(pl, p, T, out, blocks, stream) ||
           try_launch<4, true, 7, true, 4>(pl, p, T, out, blocks, stream) ||
           try_launch<4, true, 0, true, 0>(pl, p, T, out, blocks, stream);
}

}  // namespace sy
