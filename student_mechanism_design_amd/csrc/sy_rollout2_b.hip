// sy_rollout2_b.hip — rollout2 instances: boards of 129..256 nodes
// (instance list generated once; add an instance here AND in sy_dispatch.cpp::plan_rollout)
#include "sy_rollout_legacy.hpp"

namespace sy {

template <int NR, bool REC, int PT, bool POL>
static bool try_launch(const RolloutPlan& pl, const EngineParams& p, int T, const sy_rollout_buffers& out, int blocks, hipStream_t stream) {
    if (pl.nr != NR || pl.rec != REC || pl.pt != PT || pl.pol != POL) return false;
    hipLaunchKernelGGL((rollout2_kernel<NR, REC, PT, POL>), dim3(blocks), dim3(pl.threads), pl.lds, stream, p, T, out);
    return true;
}

bool launch_r2_b(const RolloutPlan& pl, const EngineParams& p, int T, const sy_rollout_buffers& out, int blocks, hipStream_t stream) {
    return try_launch<4, true, 0, false>(pl, p, T, out, blocks, stream) ||
           try_launch<4, true, 2, false>(pl, p, T, out, blocks, stream) ||
           try_launch<4, true, 4, false>(pl, p, T, out, blocks, stream) ||
           try_launch<4, true, 5, false>(pl, p, T, out, blocks, stream) ||
           try_launch<4, true, 6, false>(pl, p, T, out, blocks, stream) ||
           try_launch<4, false, 0, false>(pl, p, T, out, blocks, stream) ||
           try_launch<4, false, 2, false>(pl, p, T, out, blocks, stream) ||
           try_launch<4, false, 4, false>(pl, p, T, out, blocks, stream) ||
           try_launch<4, false, 5, false>(pl, p, T, out, blocks, stream) ||
           try_launch<4, false, 6, false>(pl, p, T, out, blocks, stream) ||
           try_launch<4, true, 4, true>(pl, p, T, out, blocks, stream) ||
           try_launch<4, true, 0, true>(pl, p, T, out, blocks, stream);
}

}  // namespace sy
