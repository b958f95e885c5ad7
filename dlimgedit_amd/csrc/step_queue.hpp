// Which lane takes which requests of the device-step queue (dlimg_amd_encode_and_mask): pure host logic, no HIP, so that
// it can be tested without a GPU (tests/test_step_queue.py through dlimg_amd_test_plan_steps).
//   - while requests arrive: one pass of `width` requests at a time to the lane with the fewest passes in flight (ties: the
//     lane after the one used last), as long as that lane has fewer than `depth` passes waiting;
//   - at a synchronisation point (`all`): every waiting request, dealt so that the images in flight per lane end up level,
//     in passes of at most `width` images whose sizes differ by at most one on a lane, launched round by round over the lanes.
#pragma once

#include <algorithm>
#include <cstddef>
#include <vector>

namespace dlimg {

struct StepPlanPass { int lane; int images; };

struct StepQueueState {
    std::vector<int> passes_in_flight;     // per lane
    std::vector<int> images_in_flight;     // per lane
    int cursor = 0;                        // lane after the one used last
};

// Plans the passes to launch now for `pending` waiting requests; updates `st` as if they had been launched and returns
// them in launch order.  What is not planned (pending - sum of images) stays queued.
inline std::vector<StepPlanPass> plan_device_steps(StepQueueState& st, int pending, int width, int depth, bool all) {
    std::vector<StepPlanPass> plan;
    const int lanes = (int)st.passes_in_flight.size();
    if (lanes <= 0 || pending <= 0) return plan;
    width = std::max(1, width);
    auto least = [&](std::vector<int> const& load) {
        int best = -1;
        for (int i = 0; i < lanes; ++i) {
            const int l = (st.cursor + i) % lanes;
            if (best < 0 || load[l] < load[best]) best = l;
        }
        st.cursor = (best + 1) % lanes;
        return best;
    };
    while (pending >= width) {
        const int saved = st.cursor;
        const int best = least(st.passes_in_flight);
        if (st.passes_in_flight[best] >= depth) {
            st.cursor = saved;              // nothing was launched: the turn is not used up
            break;
        }
        plan.push_back({best, width});
        ++st.passes_in_flight[best];
        st.images_in_flight[best] += width;
        pending -= width;
    }
    if (all && pending > 0) {
        std::vector<int> share(lanes, 0), load = st.images_in_flight;
        for (int i = 0; i < pending; ++i) {
            const int best = least(load);
            ++load[best];
            ++share[best];
        }
        std::vector<int> passes(lanes);
        int rounds = 0;
        for (int l = 0; l < lanes; ++l) {
            passes[l] = (share[l] + width - 1) / width;
            rounds = std::max(rounds, passes[l]);
        }
        for (int r = 0; r < rounds; ++r)
            for (int l = 0; l < lanes; ++l) {
                if (r >= passes[l]) continue;
                const int n = share[l] / passes[l] + (r < share[l] % passes[l] ? 1 : 0);
                plan.push_back({l, n});
                ++st.passes_in_flight[l];
                st.images_in_flight[l] += n;
            }
    }
    return plan;
}

}  // namespace dlimg
