// sy_rollout3_q.hip — rollout3 instances with the learned policy, boards of up to 128 nodes (one half-wave instance per police count, wide enough for any row)
// (add an instance here AND in sy_dispatch.hip::plan_rollout)
#include "sy_rollout3.hpp"

namespace sy {

template <int NR, bool REC, int PT, bool POL, int HS>
static bool try_launch(const RolloutPlan& pl, const EngineParams& p, int T, const sy_rollout_buffers& out, int blocks, hipStream_t stream) {
    if (pl.nr != NR || pl.rec != REC || pl.pt != PT || pl.pol != POL || pl.hs != HS) return false;
    hipLaunchKernelGGL((rollout3_kernel<NR, REC, PT, POL, HS>), dim3(blocks), dim3(pl.threads), pl.lds, stream, p, T, out);
    return true;
}

bool launch_r3_q(const RolloutPlan& pl, const EngineParams& p, int T, const sy_rollout_buffers& out, int blocks, hipStream_t stream) {
    return try_launch<1, true, 2, true, 2>(pl, p, T, out, blocks, stream) ||
           try_launch<1, true, 4, true, 3>(pl, p, T, out, blocks, stream) ||
           try_launch<1, true, 5, true, 4>(pl, p, T, out, blocks, stream) ||
           try_launch<1, true, 6, true, 4>(pl, p, T, out, blocks, stream) ||
           try_launch<1, true, 7, true, 4>(pl, p, T, out, blocks, stream) ||
           try_launch<1, true, 0, true, 0>(pl, p, T, out, blocks, stream) ||
           try_launch<2, true, 2, true, 2>(pl, p, T, out, blocks, stream) ||
           try_launch<2, true, 4, true, 3>(pl, p, T, out, blocks, stream) ||
           try_launch<2, true, 5, true, 4>(pl, p, T, out, blocks, stream) ||
           try_launch<2, true, 6, true, 4>(pl, p, T, out, blocks, stream) ||
           try_launch<2, true, 7, true, 4>(pl, p, T, out, blocks, stream) ||
           try_launch<2, true, 0, true, 0>(pl, p, T, out, blocks, stream);
}

}  // namespace sy
