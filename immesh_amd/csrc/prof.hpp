// Optional per-kernel timing with HIP events recorded on the stream each kernel is launched on (bench.py's roofline
// leg reads it through immesh_profile_read).  Off by default: a disabled profiler costs one thread-local load per launch.
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include <cstring>

struct KProf {
    bool on = false;
    struct Rec { int id; hipEvent_t a, b; };
    std::vector<std::string> names;
    std::vector<double> ms;
    std::vector<long long> cnt;
    std::vector<Rec> pending;
    std::vector<hipEvent_t> pool;

    int id_of(const char* n0) {
        std::string n(n0);   // (instantiations of one kernel template report under the template's name: residual_persistent_kernel<true> / <false>)
        const size_t lt = n.find('<');
        if (lt != std::string::npos) n.resize(lt);
        for (size_t i = 0; i < names.size(); i++) if (names[i] == n) return (int)i;
        names.emplace_back(n); ms.push_back(0.0); cnt.push_back(0);
        return (int)names.size() - 1;
    }
    hipEvent_t get_event() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e; (void)hipEventCreate(&e); return e;
    }
    void flush() {  // caller has synchronised the stream
        for (const Rec& r : pending) {
            float t = 0.f;
            if (hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) { ms[r.id] += (double)t; cnt[r.id]++; }
            pool.push_back(r.a); pool.push_back(r.b);
        }
        pending.clear();
    }
    void reset() { for (auto& v : ms) v = 0; for (auto& v : cnt) v = 0; }
    ~KProf() { for (const Rec& r : pending) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); } for (hipEvent_t e : pool) (void)hipEventDestroy(e); }
};

extern thread_local KProf* g_kprof;
// The pipeline is latency-bound: the stream is usually idle when a kernel is submitted, so a start event would execute at once and the
// measured interval would include the host's submission latency (several us, more than some kernels run).  A short spin kernel ahead of
// the start event keeps the queue busy until start event, kernel and stop event are all enqueued; they then execute back to back.
void kprof_spin(hipStream_t s);

struct KProfScope {
    KProf* p; hipStream_t s; KProf::Rec r;
    KProfScope(const char* name, hipStream_t stream) : p(g_kprof), s(stream) {
        if (p && p->on) { r.id = p->id_of(name); r.a = p->get_event(); r.b = p->get_event(); kprof_spin(s); (void)hipEventRecord(r.a, s); } else p = nullptr;
    }
    ~KProfScope() { if (p) { (void)hipEventRecord(r.b, s); p->pending.push_back(r); } }
};

// launch `kernel` on `stream`; timed when the calling thread's context has profiling enabled
#define KLAUNCH(kernel, grid, block, shmem, stream, ...)                              \
    do {                                                                              \
        KProfScope _kps(#kernel, stream);                                             \
        hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);          \
    } while (0)
// time a library call (rocPRIM sort / scan) that launches several kernels under one name
#define KTIMED(name, stream, call)                                                    \
    do {                                                                              \
        KProfScope _kps(name, stream);                                                \
        call;                                                                         \
    } while (0)
