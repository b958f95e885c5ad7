// ThreadSanitizer harness for csrc/lane_worker.hpp and the hand-over protocols built on it (tests/test_sanitizers.py):
//   1. several producer threads post to several workers while other threads drain them;
//   2. the step queue's ticket protocol (environment.hpp, StepTicket): the worker writes `done`, then state (release); the
//      planner reads state (acquire), then `done`;
//   3. the batch call's promise hand-over (segmentation.cpp, process_batch): the task shares the promise with the caller.
#include "lane_worker.hpp"

#include <atomic>
#include <cstdio>
#include <future>
#include <memory>
#include <vector>

using dlimg::LaneWorker;

struct Ticket { std::atomic<int> state{0}; void* done = nullptr; };

int main() {
    long total = 0;
    {   // 1
        std::vector<std::unique_ptr<LaneWorker>> workers;
        for (int i = 0; i < 4; ++i) workers.push_back(std::make_unique<LaneWorker>());
        std::vector<long> sums(4, 0);                       // sums[w] is only touched by worker w's thread
        std::vector<std::thread> producers;
        for (int p = 0; p < 3; ++p)
            producers.emplace_back([&, p] {
                for (int i = 0; i < 2000; ++i) {
                    const int w = (i + p) % 4;
                    workers[w]->post([&sums, w, i] { sums[w] += i; });
                    if (i % 257 == 0) workers[(w + 1) % 4]->drain();
                }
            });
        for (auto& t : producers) t.join();
        for (auto& w : workers) w->drain();
        for (long s : sums) total += s;
        if (total != 3L * (1999L * 2000L / 2)) { std::printf("lost tasks: %ld\n", total); return 1; }
    }
    {   // 2
        LaneWorker worker;
        static int payload[64];
        std::vector<std::shared_ptr<Ticket>> tickets;
        for (int i = 0; i < 64; ++i) {
            auto t = std::make_shared<Ticket>();
            tickets.push_back(t);
            worker.post([t, i] {
                payload[i] = i * 7;
                t->done = &payload[i];
                t->state.store(1, std::memory_order_release);
            });
        }
        size_t retired = 0;
        while (retired < tickets.size()) {                  // the planner polls the oldest ticket, as retire_device_steps does
            Ticket& t = *tickets[retired];
            if (t.state.load(std::memory_order_acquire) == 0) { std::this_thread::yield(); continue; }
            if (*static_cast<int*>(t.done) != (int)retired * 7) { std::printf("ticket %zu: wrong payload\n", retired); return 1; }
            t.done = nullptr;
            ++retired;
        }
    }
    {   // 3
        LaneWorker a, b;
        for (int round = 0; round < 200; ++round) {
            struct Handed { std::promise<int> result; std::future<int> answer; };
            std::vector<std::shared_ptr<Handed>> handed;
            int frame_local = round;                        // the tasks refer to the caller's frame, as run_chunk does
            for (int i = 0; i < 4; ++i) {
                auto h = std::make_shared<Handed>();
                h->answer = h->result.get_future();
                handed.push_back(h);
                (i & 1 ? a : b).post([h, &frame_local, i] { h->result.set_value(frame_local * 4 + i); });
            }
            for (int i = 0; i < 4; ++i)
                if (handed[i]->answer.get() != round * 4 + i) { std::printf("round %d: wrong answer\n", round); return 1; }
        }
    }
    std::printf("ok\n");
    return 0;
}
