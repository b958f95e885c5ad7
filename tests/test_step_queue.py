"""Host logic of the device-step queue behind dlimg_amd_encode_and_mask (csrc/step_queue.hpp), through the library's
GPU-free hook dlimg_amd_test_plan_steps: which lane takes which requests while they arrive and at a synchronisation point."""
import pytest

from dlimgedit_amd import api


def burst(n_requests, lanes=4, width=2, depth=2):
    """n single-image requests arriving one by one, then dlimg_amd_synchronize: returns the passes per lane."""
    passes, images, cursor, pending = [0] * lanes, [0] * lanes, 0, 0
    per_lane = [[] for _ in range(lanes)]
    order = []
    for _ in range(n_requests):
        pending += 1
        plan, passes, images, cursor = api.ext.plan_steps(passes, images, cursor, pending, width, depth, False)
        for lane, n in plan:
            per_lane[lane].append(n)
            order.append(lane)
            pending -= n
    plan, passes, images, cursor = api.ext.plan_steps(passes, images, cursor, pending, width, depth, True)
    for lane, n in plan:
        per_lane[lane].append(n)
        order.append(lane)
        pending -= n
    assert pending == 0
    assert images == [sum(p) for p in per_lane] and passes == [len(p) for p in per_lane]
    return per_lane, order


def test_official_block_of_20_requests_ends_level():
    per_lane, order = burst(20)
    assert [sum(p) for p in per_lane] == [5, 5, 5, 5]               # not 6 / 6 / 4 / 4
    assert all(p == [2, 2, 1] for p in per_lane)
    assert order[:8] == [0, 1, 2, 3, 0, 1, 2, 3]                     # the lanes take turns while requests arrive


@pytest.mark.parametrize("n", [1, 2, 3, 5, 7, 8, 9, 16, 17, 23, 40, 41])
@pytest.mark.parametrize("lanes,width,depth", [(4, 2, 2), (3, 2, 2), (4, 1, 2), (4, 3, 1), (4, 4, 2), (1, 2, 2)])
def test_every_request_runs_once_in_bounded_passes_and_lanes_end_within_one_pass(n, lanes, width, depth):
    per_lane, _ = burst(n, lanes, width, depth)
    totals = [sum(p) for p in per_lane]
    assert sum(totals) == n
    assert all(1 <= size <= width for p in per_lane for size in p)
    # what was dealt at the synchronisation point levels the lanes: they differ by less than one early pass
    assert max(totals) - min(totals) <= width


def test_depth_bounds_what_waits_on_a_lane_before_the_synchronisation_point():
    passes, images, cursor, pending = [0] * 4, [0] * 4, 0, 0
    launched = 0
    for _ in range(40):
        pending += 1
        plan, passes, images, cursor = api.ext.plan_steps(passes, images, cursor, pending, 2, 2, False)
        launched += sum(n for _, n in plan)
        pending -= sum(n for _, n in plan)
    assert passes == [2, 2, 2, 2] and launched == 16 and pending == 24


def test_a_lane_that_has_drained_is_preferred():
    # lanes 0, 1, 3 still have two passes in flight, lane 2 none: the next pass goes there whatever the cursor says
    plan, passes, images, cursor = api.ext.plan_steps([2, 2, 0, 2], [4, 4, 0, 4], 0, 2, 2, 2, False)
    assert plan == [(2, 2)] and passes == [2, 2, 1, 2] and cursor == 3


def test_ties_take_turns_when_completion_cannot_be_observed():
    # every pass looks finished at once (what rocprofv3's kernel trace does to hipEventQuery): plain round robin
    cursor, seen = 0, []
    for _ in range(8):
        plan, _, _, cursor = api.ext.plan_steps([0] * 4, [0] * 4, cursor, 2, 2, 2, False)
        seen.append(plan[0][0])
    assert seen == [0, 1, 2, 3, 0, 1, 2, 3]


def test_lane_worker_runs_tasks_in_order_and_drain_waits():
    """The lanes' enqueue threads (csrc/environment.hpp, LaneWorker): tasks run one at a time in posting order, drain()
    returns only when nothing is queued or running, and the destructor finishes what was posted after the drain."""
    from dlimgedit_amd import api
    assert api.ext.test_lane_worker(0) == [0]
    assert api.ext.test_lane_worker(1, 200) == [0, 1]
    assert api.ext.test_lane_worker(40, 50) == list(range(41))
    assert api.ext.test_lane_worker(500, 0) == list(range(501))
