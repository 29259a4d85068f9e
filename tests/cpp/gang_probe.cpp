// The thread rendezvous of GpuPathIntegrator::RenderAllDevices (csrc/host/device_gang.h) with STUB backends, on the CPU: what the
// gang does when a device cannot be selected, a communicator cannot be made, a thread never arrives, a render fails — every case
// must return false within the deadline with every thread joined, and nobody may be inside the (stub) collective when it does.
//   gang_probe <case> <n> <visible> <timeout_s>     prints one line of JSON; exit code 0 = the gang behaved
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../pbrt-v3-iile_amd/csrc/host/device_gang.h"

using Clock = std::chrono::steady_clock;

struct StubComm {
    int rank;
};

struct Stub {
    std::string fault;   // none | select | create | absent | slow_create | run
    int fault_rank = 0;
    double create_timeout_s = 0.5;    // what iile_dist_create_deadline would wait for a missing rank
    std::atomic<int> selected{0}, created{0}, aborted{0}, destroyed{0}, ran{0}, in_collective{0}, collective_entered_short{0};
    int n = 0;
    std::atomic<int> joined{0};   // ranks that have called CreateComm (the stub's "ncclCommInitRank": completes when all n are in)

    bool SelectDevice(int r) {
        if (fault == "absent" && r == fault_rank) std::this_thread::sleep_for(std::chrono::duration<double>(3600));   // (never reached: see main)
        if (fault == "select" && r == fault_rank) return false;
        ++selected;
        return true;
    }
    void *CreateComm(int r, int nranks) {
        if (fault == "create" && r == fault_rank) return nullptr;
        // a collective set-up: waits for all ranks, at most create_timeout_s (the deadline communicator)
        ++joined;
        const Clock::time_point t0 = Clock::now();
        while (joined.load() < nranks) {
            if (std::chrono::duration<double>(Clock::now() - t0).count() > create_timeout_s) return nullptr;
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
        if (fault == "slow_create" && r == fault_rank) std::this_thread::sleep_for(std::chrono::duration<double>(0.8 * create_timeout_s));
        ++created;
        return new StubComm{r};
    }
    bool Run(int r, void *comm) {
        // the frame's collective: every rank must be here, or the job would hang — count who enters with fewer than n created
        if (created.load() < n) ++collective_entered_short;
        ++in_collective;
        ++ran;
        --in_collective;
        return !(fault == "run" && r == fault_rank) && static_cast<StubComm *>(comm)->rank == r;
    }
    void DestroyComm(void *comm) {
        ++destroyed;
        delete static_cast<StubComm *>(comm);
    }
    void AbortComm(void *comm) {
        ++aborted;
        delete static_cast<StubComm *>(comm);
    }
};

int main(int argc, char **argv) {
    if (argc != 5) return 2;
    Stub be;
    be.fault = argv[1];
    const size_t colon = be.fault.find(':');
    if (colon != std::string::npos) be.fault_rank = atoi(be.fault.c_str() + colon + 1), be.fault = be.fault.substr(0, colon);
    const int n = atoi(argv[2]), visible = atoi(argv[3]);
    const double timeout_s = atof(argv[4]);
    be.n = n;
    be.create_timeout_s = timeout_s;
    std::string why;
    const Clock::time_point t0 = Clock::now();
    bool ok;
    if (be.fault == "vote_absent") {
        // GangVote alone: n - 1 threads vote yes, one never comes: everybody must get `false` after the deadline, later votes at once
        iile::GangVote vote(n, timeout_s);
        std::vector<std::thread> th;
        std::atomic<int> yes{0};
        for (int r = 0; r < n - 1; ++r) th.emplace_back([&] { if (vote.Vote(true)) ++yes; if (vote.Vote(true)) ++yes; });
        for (auto &t : th) t.join();
        ok = yes.load() == 0 && vote.broken();
        why = vote.why();
        const double dt = std::chrono::duration<double>(Clock::now() - t0).count();
        printf("{\"case\": \"vote_absent\", \"ok\": %s, \"seconds\": %.3f, \"why\": \"%s\"}\n", ok ? "true" : "false", dt, why.c_str());
        return ok && dt < 2.0 * timeout_s + 1.0 ? 0 : 1;
    }
    ok = iile::RunGang(be, n, visible, timeout_s, &why);
    const double dt = std::chrono::duration<double>(Clock::now() - t0).count();
    printf("{\"case\": \"%s\", \"n\": %d, \"visible\": %d, \"returned\": %s, \"seconds\": %.3f, \"selected\": %d, \"created\": %d, \"ran\": %d, \"destroyed\": %d, "
           "\"aborted\": %d, \"collective_entered_short\": %d, \"why\": \"%s\"}\n",
           argv[1], n, visible, ok ? "true" : "false", dt, be.selected.load(), be.created.load(), be.ran.load(), be.destroyed.load(), be.aborted.load(),
           be.collective_entered_short.load(), why.c_str());
    return 0;
}
