// HipError.cc -- error channel of the drop-in classes (include/orbhip/hiperror.h): no exceptions by default.
#include "hiperror.h"

#include <atomic>
#include <cstdio>
#include <mutex>
#include <set>
#include <string>
#ifdef ORBHIP_THROW
#include <stdexcept>
#endif

namespace ORB_SLAM2
{
namespace
{
thread_local std::string t_last;
std::atomic<unsigned long> g_count{0};
std::mutex g_seen_mutex;
std::set<std::string> g_seen;   // call sites that have already been logged
}  // namespace

const char *OrbHipLastError() { return t_last.c_str(); }
unsigned long OrbHipErrorCount() { return g_count.load(); }

namespace hipdetail
{
bool Fail(const char *who, const char *msg)
{
    t_last = std::string(who ? who : "orbhip") + ": " + (msg && *msg ? msg : "unknown error");
    g_count++;
#ifdef ORBHIP_THROW
    throw std::runtime_error(t_last);
#else
    bool first;
    {
        std::lock_guard<std::mutex> lk(g_seen_mutex);
        first = g_seen.insert(who ? who : "").second;
    }
    if (first)
        fprintf(stderr, "[orbhip] %s -- the call returns an empty result (further failures of this call are counted, not "
                        "printed; OrbHipLastError() / OrbHipErrorCount())\n", t_last.c_str());
    return false;
#endif
}
}  // namespace hipdetail
}  // namespace ORB_SLAM2
