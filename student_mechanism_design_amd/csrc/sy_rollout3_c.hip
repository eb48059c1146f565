// sy_rollout3_c.hip — rollout3 instances: boards of up to 64 nodes
// (instance list generated once; add an instance here AND in sy_dispatch.cpp::plan_rollout)
#include "sy_rollout3.hpp"

namespace sy {

template <int NR, bool REC, int PT, bool POL, int HS>
static bool try_launch(const RolloutPlan& pl, const EngineParams& p, int T, const sy_rollout_buffers& out, int blocks, hipStream_t stream) {
    if (pl.nr != NR || pl.rec != REC || pl.pt != PT || pl.pol != POL || pl.hs != HS) return false;
    hipLaunchKernelGGL((rollout3_kernel<NR, REC, PT, POL, HS>), dim3(blocks), dim3(pl.threads), pl.lds, stream, p, T, out);
    return true;
}

bool launch_r3_c(const RolloutPlan& pl, const EngineParams& p, int T, const sy_rollout_buffers& out, int blocks, hipStream_t stream) {
    return try_launch<1, true, 0, false, 0>(pl, p, T, out, blocks, stream) ||
           try_launch<1, true, 2, false, 0>(pl, p, T, out, blocks, stream) ||
           try_launch<1, true, 4, false, 0>(pl, p, T, out, blocks, stream) ||
           try_launch<1, true, 5, false, 0>(pl, p, T, out, blocks, stream) ||
           try_launch<1, true, 6, false, 0>(pl, p, T, out, blocks, stream) ||
           try_launch<1, true, 2, false, 2>(pl, p, T, out, blocks, stream) ||
           try_launch<1, true, 4, false, 2>(pl, p, T, out, blocks, stream) ||
           try_launch<1, false, 0, false, 0>(pl, p, T, out, blocks, stream) ||
           try_launch<1, false, 2, false, 0>(pl, p, T, out, blocks, stream) ||
           try_launch<1, false, 4, false, 0>(pl, p, T, out, blocks, stream) ||
           try_launch<1, false, 5, false, 0>(pl, p, T, out, blocks, stream) ||
           try_launch<1, false, 6, false, 0>(pl, p, T, out, blocks, stream) ||
           try_launch<1, false, 2, false, 2>(pl, p, T, out, blocks, stream) ||
           try_launch<1, false, 4, false, 2>(pl, p, T, out, blocks, stream);
}

}  // namespace sy
