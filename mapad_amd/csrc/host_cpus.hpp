// host_cpus.hpp — how many CPUs this process may really use.
// A container often shows every CPU of the machine (the GPU boxes of this pool: 256) and limits the CPU TIME of its cgroup (cpu.max = 16 CPUs' worth).  Threads
// beyond that share do not add throughput: the cgroup burns its quota of a 100 ms period in a fraction of it and is then stopped as a whole until the next period
// — every thread of the process, the one that feeds the GPU included (round 4: 943 throttled periods, 207 s of throttled thread time in one 125 s run).  Every
// thread pool of the host side (host tail, post-search strings, index preparation, the command line's reader and writer) sizes itself by this number.
#pragma once
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <thread>

namespace mapad {
namespace host {

inline unsigned cpu_share() {
    static const unsigned share = [] {
        if (const char* e = std::getenv("MAPAD_HOST_CPUS")) { const unsigned v = (unsigned)std::strtoul(e, nullptr, 10); if (v) return v; }
        unsigned n = std::thread::hardware_concurrency();
        if (!n) n = 8;
        auto read2 = [](const char* path, double& a, double& b) -> bool {
            FILE* f = std::fopen(path, "r");
            if (!f) return false;
            char q[64] = {0}, per[64] = {0};
            const int k = std::fscanf(f, "%63s %63s", q, per);
            std::fclose(f);
            if (k < 1 || q[0] == 'm' /* "max" */) return false;
            a = std::atof(q); b = k == 2 ? std::atof(per) : 0.0;
            return a > 0;
        };
        double quota = 0, period = 0;
        if (read2("/sys/fs/cgroup/cpu.max", quota, period) && period > 0) n = std::min<unsigned>(n, (unsigned)std::max(1.0, quota / period + 0.5));
        else {
            double q1 = 0, p1 = 0, dummy = 0;
            if (read2("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", q1, dummy) && read2("/sys/fs/cgroup/cpu/cpu.cfs_period_us", p1, dummy) && p1 > 0)
                n = std::min<unsigned>(n, (unsigned)std::max(1.0, q1 / p1 + 0.5));
        }
        return std::max(1u, n);
    }();
    return share;
}

}  // namespace host
}  // namespace mapad
