// sy_rollout1_a.hip — rollout_kernel instances (odd block sizes): boards of up to 256 nodes
// (instance list generated once; add an instance here AND in sy_dispatch.cpp::plan_rollout)
#include "sy_rollout_legacy.hpp"

namespace sy {

template <int NR, bool REC, int PT>
static bool try_launch(const RolloutPlan& pl, const EngineParams& p, int T, const sy_rollout_buffers& out, int blocks, hipStream_t stream) {
    if (pl.nr != NR || pl.rec != REC || pl.pt != PT) return false;
    hipLaunchKernelGGL((rollout_kernel<NR, REC, PT>), dim3(blocks), dim3(pl.threads), pl.lds, stream, p, T, out);
    return true;
}

bool launch_r1_a(const RolloutPlan& pl, const EngineParams& p, int T, const sy_rollout_buffers& out, int blocks, hipStream_t stream) {
    return try_launch<1, true, 0>(pl, p, T, out, blocks, stream) ||
           try_launch<1, true, 2>(pl, p, T, out, blocks, stream) ||
           try_launch<1, true, 4>(pl, p, T, out, blocks, stream) ||
           try_launch<1, true, 5>(pl, p, T, out, blocks, stream) ||
           try_launch<1, true, 6>(pl, p, T, out, blocks, stream) ||
           try_launch<1, false, 0>(pl, p, T, out, blocks, stream) ||
           try_launch<1, false, 2>(pl, p, T, out, blocks, stream) ||
           try_launch<1, false, 4>(pl, p, T, out, blocks, stream) ||
           try_launch<1, false, 5>(pl, p, T, out, blocks, stream) ||
           try_launch<1, false, 6>(pl, p, T, out, blocks, stream) ||
           try_launch<2, true, 0>(pl, p, T, out, blocks, stream) ||
           try_launch<2, true, 2>(pl, p, T, out, blocks, stream) ||
           try_launch<2, true, 4>(pl, p, T, out, blocks, stream) ||
           try_launch<2, true, 5>(pl, p, T, out, blocks, stream) ||
           try_launch<2, true, 6>(pl, p, T, out, blocks, stream) ||
           try_launch<2, false, 0>(pl, p, T, out, blocks, stream) ||
           try_launch<2, false, 2>(pl, p, T, out, blocks, stream) ||
           try_launch<2, false, 4>(pl, p, T, out, blocks, stream) ||
           try_launch<2, false, 5>(pl, p, T, out, blocks, stream) ||
           try_launch<2, false, 6>(pl, p, T, out, blocks, stream) ||
           try_launch<4, true, 0>(pl, p, T, out, blocks, stream) ||
           try_launch<4, true, 2>(pl, p, T, out, blocks, stream) ||
           try_launch<4, true, 4>(pl, p, T, out, blocks, stream) ||
           try_launch<4, true, 5>(pl, p, T, out, blocks, stream) ||
           try_launch<4, true, 6>(pl, p, T, out, blocks, stream) ||
           try_launch<4, false, 0>(pl, p, T, out, blocks, stream) ||
           try_launch<4, false, 2>(pl, p, T, out, blocks, stream) ||
           try_launch<4, false, 4>(pl, p, T, out, blocks, stream) ||
           try_launch<4, false, 5>(pl, p, T, out, blocks, stream) ||
           try_launch<4, false, 6>(pl, p, T, out, blocks, stream);
}

}  // namespace sy
