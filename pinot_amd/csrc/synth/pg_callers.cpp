// Bench / test helper (libpinot_synth.so, next to the synthetic segment writer; NOT part of libpinot_gpu.so): N host threads calling
// pg_query_exec on one segment for a fixed time — the reference runs many queries at once, one worker thread per segment and query
// (BaseCombineOperator.java:97-142), and the boundary promises per-(thread, device) streams.  The threads are native so that the
// measured loop holds no interpreter: bench.py passes the entry points' addresses (no link-time dependency on the library), the
// segment and the pg_query it built; latencies come back per call.
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

namespace {
typedef int32_t (*ExecFn)(void* segment, const void* query, void** out_result);
typedef int32_t (*FreeFn)(void* result);
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace

extern "C" {

// Runs `n_threads` callers of (segment, query) for `seconds` after `warmup_s`; optionally ONE more thread loops (bg_segment, bg_query) the
// whole time (a long scan in the background).  lat_ms receives up to `cap` per-call latencies of the timed window (callers interleaved),
// *n_done the calls completed in it, *bg_done the background calls; returns the window's length in seconds, or -1 with the first failing
// status in *first_error.
double pgs_concurrent_callers(void* exec_fn, void* free_fn, void* segment, const void* query, int n_threads, double warmup_s, double seconds,
                              float* lat_ms, int64_t cap, int64_t* n_done, void* bg_segment, const void* bg_query, int64_t* bg_done,
                              int32_t* first_error) {
  const ExecFn exec = reinterpret_cast<ExecFn>(exec_fn);
  const FreeFn release = reinterpret_cast<FreeFn>(free_fn);
  std::atomic<int> phase{0};   // 0 warm-up, 1 timed, 2 stop
  std::atomic<int32_t> err{0};
  std::atomic<int64_t> bg_calls{0};
  std::vector<std::vector<float>> per((size_t)n_threads);
  std::vector<std::atomic<int>> warm((size_t)n_threads);   // calls a thread has completed (its stream, events and pinned block exist after the first)
  for (auto& w : warm) w.store(0);
  std::vector<std::thread> threads;
  for (int t = 0; t < n_threads; t++)
    threads.emplace_back([&, t] {
      per[(size_t)t].reserve(1 << 16);
      while (phase.load(std::memory_order_acquire) != 2) {
        const double t0 = now_s();
        void* r = nullptr;
        const int32_t st = exec(segment, query, &r);
        const double t1 = now_s();
        if (st < 0) { int32_t z = 0; err.compare_exchange_strong(z, st); break; }
        release(r);
        warm[(size_t)t].fetch_add(1, std::memory_order_release);
        if (phase.load(std::memory_order_acquire) == 1) per[(size_t)t].push_back((float)((t1 - t0) * 1e3));
      }
    });
  std::thread bg;
  if (bg_segment && bg_query)
    bg = std::thread([&] {
      while (phase.load(std::memory_order_acquire) != 2) {
        void* r = nullptr;
        const int32_t st = exec(bg_segment, bg_query, &r);
        if (st < 0) { int32_t z = 0; err.compare_exchange_strong(z, st); break; }
        release(r);
        if (phase.load(std::memory_order_acquire) == 1) bg_calls++;
      }
    });
  // warm-up: at least `warmup_s`, and until every caller has three calls behind it (a new thread's first call creates its stream, its
  // events and its page-locked block — tens of milliseconds when 64 threads do so at once) or 20 s have passed
  std::this_thread::sleep_for(std::chrono::duration<double>(warmup_s));
  for (const double give_up = now_s() + 20.0; now_s() < give_up && !err.load();) {
    bool all = true;
    for (auto& w : warm) all = all && w.load(std::memory_order_acquire) >= 3;
    if (all) break;
    std::this_thread::sleep_for(std::chrono::milliseconds(2));
  }
  const double w0 = now_s();
  phase.store(1, std::memory_order_release);
  std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
  phase.store(2, std::memory_order_release);
  const double w1 = now_s();
  for (auto& th : threads) th.join();
  if (bg.joinable()) bg.join();
  int64_t n = 0, written = 0;
  for (auto& v : per) n += (int64_t)v.size();
  // interleave the callers' samples so that a truncated copy still holds every caller
  for (size_t i = 0; written < cap; i++) {
    bool any = false;
    for (auto& v : per)
      if (i < v.size() && written < cap) { lat_ms[written++] = v[i]; any = true; }
    if (!any) break;
  }
  if (n_done) *n_done = n;
  if (bg_done) *bg_done = bg_calls.load();
  if (first_error) *first_error = err.load();
  return err.load() ? -1.0 : w1 - w0;
}

}  // extern "C"
