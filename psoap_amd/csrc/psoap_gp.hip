// psoap_gp.hip -- C ABI (include/psoap_gp.h) over the gfx950 kernels.
// Host side only: device memory, streams, launch sequencing, error mapping.
#include "../../include/psoap_gp.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <stddef.h>
#include <string.h>

#include <errno.h>
#include <dirent.h>
#include <fcntl.h>
#include <time.h>
#include <sys/file.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "chol_kernels.hpp"
#include "dag_kernel.hpp"
#include "solo_kernel.hpp"
#include "fill_kernels.hpp"
#include "orbit_kernels.hpp"
#include "predict_kernels.hpp"
#include "calibrate_kernels.hpp"

using namespace psoap;

static thread_local std::string g_err;

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            char _buf[512];                                                                        \
            snprintf(_buf, sizeof _buf, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                     __LINE__);                                                                    \
            g_err = _buf;                                                                          \
            return 1;                                                                              \
        }                                                                                          \
    } while (0)

#define FAIL(msg)                \
    do {                         \
        g_err = std::string(msg); \
        return 2;                \
    } while (0)

static const int MAX_GROUPS = 8;

// ---- several processes on one device ------------------------------------------------------------------------------
// The reference forks one worker PROCESS per chunk (psoap/sample_parallel.py:258-278); with more chunks than GPUs several
// of them share a device.  Two things then go wrong with persistent kernels:
//   * two persistent launches of different processes on the device at once starve each other (every workgroup that holds
//     a ticket is assumed to run): bounded waits time out, values come back wrong;
//   * the device suspends running workgroups (compute wave save / restore) when more than eight processes or too many
//     hardware queues share it, and resumes them on other compute units: see dag_where() in dag_kernel.hpp.
// What the library does (round 5):
//   1. an advisory per-device lock between the processes of ONE user -- flock on <dir>/gpu_<PCI bus id>.lock, <dir> =
//      $PSOAP_LOCK_DIR, else $XDG_RUNTIME_DIR/psoap, else /tmp/psoap-<uid> (0700, owner checked, never followed through a
//      symlink) -- held from an upload to the fetch of the evaluation that reads it, counted within a process, with a
//      time-out (PSOAP_DEVICE_LOCK_TIMEOUT_S, default 300) that ends in an error naming the holder instead of a hang;
//   2. a count of the cooperating processes per device (slot files beside the lock, locked for the life of the process,
//      re-counted a few times per second): share_procs();
//   3. every persistent launch reports workgroups that MOVED while they ran (DagCtl::pad[3]); a tainted evaluation is
//      run again (PSOAP_SHARE_RETRIES, default 3, same task list: bit-identical results), then evaluated by the staged
//      path (chol_kernels.hpp: kernel boundaries instead of hand-offs inside a kernel -- immune to both failures);
//   4. where the persistent kernel cannot be made safe -- several processes on the device WITHOUT the lock
//      (PSOAP_DEVICE_LOCK=0), or more of them than PSOAP_SHARE_DAG_MAX (8: the process contexts the device keeps mapped) --
//      evaluations take the staged path from the start, and take it WITHOUT the lock: kernels that synchronise at their
//      boundaries only have nothing to keep apart, and the device interleaves them (16 processes, N = 6000: 143
//      evaluations per second in all, against 91 with persistent launches taking turns under the lock -- one of 38,400 of
//      those still wrong despite the retries -- and 16 with staged evaluations taking turns: profiles/r5_share_*.txt).
//      Streams (resident launches) are refused in that regime.
// PSOAP_SHARE_POLICY=dag|staged pins the path whatever the count (experiments, tools/shared_gpu_probe.py).
struct DeviceLock {
    std::mutex mu;                    // guards the fields below (one per device: a wait on one device never blocks another)
    std::condition_variable cv;
    int fd = -1;
    int refs = 0;
    bool held = false, acquiring = false, broken = false;
    pid_t pid = 0;
    std::string path;
};
static std::mutex g_devlock_mu;        // guards the maps only
static std::map<int, DeviceLock*> g_devlocks;

static bool device_lock_enabled()
{
    static const bool on = !(getenv("PSOAP_DEVICE_LOCK") && getenv("PSOAP_DEVICE_LOCK")[0] == '0');
    return on;
}
static double device_lock_timeout_s()
{
    const char* e = getenv("PSOAP_DEVICE_LOCK_TIMEOUT_S");
    return (e && atof(e) > 0.0) ? atof(e) : 300.0;
}

// process-wide counters of what sharing cost (psoap_share_stats)
struct ShareStats {
    std::atomic<long long> dag_launches{0}, tainted{0}, retries{0}, staged_fallbacks{0}, staged_policy{0}, moved_tasks{0},
        moved_xcd{0}, lock_acquisitions{0}, lock_wait_us{0}, stream_resubmits{0};
};
static ShareStats g_share;

// The per-user directory of the lock and slot files; empty when none can be had (the caller then runs unserialised and
// says so once).  Created 0700; refused when it is a symlink, not a directory, or somebody else's.
static const std::string& share_dir()
{
    static std::string dir;
    static std::once_flag once;
    std::call_once(once, [] {
        std::string d;
        if (const char* e = getenv("PSOAP_LOCK_DIR")) d = e;
        else if (const char* x = getenv("XDG_RUNTIME_DIR")) d = std::string(x) + "/psoap";
        else d = std::string("/tmp/psoap-") + std::to_string((long long)geteuid());
        if (mkdir(d.c_str(), 0700) != 0 && errno != EEXIST) return;
        struct stat sb;
        if (lstat(d.c_str(), &sb) != 0 || !S_ISDIR(sb.st_mode) || sb.st_uid != geteuid()) {
            fprintf(stderr, "psoap: %s is not a directory of this user: several processes on one GPU are not serialised\n", d.c_str());
            return;
        }
        dir = d;
    });
    return dir;
}

static std::string device_file(int device, const char* suffix)
{
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, device) != hipSuccess) snprintf(bus, sizeof bus, "index%d", device);
    for (char* c = bus; *c; ++c)
        if (!((*c >= '0' && *c <= '9') || (*c >= 'a' && *c <= 'z') || (*c >= 'A' && *c <= 'Z'))) *c = '_';
    return share_dir() + "/gpu_" + bus + suffix;
}

static int open_share_file(const std::string& path, bool create)
{
    return open(path.c_str(), (create ? O_CREAT : 0) | O_RDWR | O_CLOEXEC | O_NOFOLLOW, 0600);
}

// 0: the device is this process's (*took), or no lock is needed -- switched off / unavailable, or this process runs the
// staged path on a device too many processes share (share_wants_staged: kernel boundaries only, nothing to keep apart, and
// 16 processes interleaving their kernels measured 143 evaluations per second at N = 6000 against 16 taking turns);
// 2: timed out (g_err says who holds it)
static bool share_wants_staged(int device);
static int device_lock_acquire(int device, bool* took = nullptr)
{
    if (took) *took = false;
    if (!device_lock_enabled()) return 0;
    if (share_wants_staged(device)) return 0;
    DeviceLock* Lp;
    {
        std::lock_guard<std::mutex> g(g_devlock_mu);
        DeviceLock*& slot = g_devlocks[device];
        if (!slot) slot = new DeviceLock();
        Lp = slot;
    }
    DeviceLock& L = *Lp;
    std::unique_lock<std::mutex> lk(L.mu);
    if (L.pid != getpid()) {          // first use in this process (a descriptor inherited through fork() shares its lock)
        if (L.fd >= 0) (void)close(L.fd);
        L.fd = -1;
        L.refs = 0;
        L.held = L.acquiring = false;
        L.pid = getpid();
        if (!share_dir().empty()) {
            L.path = device_file(device, ".lock");
            L.fd = open_share_file(L.path, true);
        }
        if (L.fd < 0 && !L.broken) {
            L.broken = true;
            fprintf(stderr, "psoap: cannot open the device lock %s: several processes on this GPU are not serialised\n",
                    L.path.empty() ? "(no lock directory)" : L.path.c_str());
        }
    }
    if (L.fd < 0) return 0;
    ++L.refs;
    if (took) *took = true;
    if (L.held) return 0;
    if (L.acquiring) {                 // another thread of this process is at it: wait for its verdict
        L.cv.wait(lk, [&] { return !L.acquiring; });
        if (L.held) return 0;
        if (took) *took = false;
        --L.refs;
        g_err = "psoap: the device lock could not be taken (see the other thread's error)";
        return 2;
    }
    L.acquiring = true;
    const int fd = L.fd;
    lk.unlock();
    // (polled, not blocking: a wait that never ends must become an error.  20 us steps at first -- the holder's evaluation
    // takes milliseconds -- then 200 us.)
    const auto t0 = std::chrono::steady_clock::now();
    const double limit = device_lock_timeout_s();
    bool got = false;
    long long polls = 0;
    for (;;) {
        if (flock(fd, LOCK_EX | LOCK_NB) == 0) {
            got = true;
            break;
        }
        if (errno != EWOULDBLOCK && errno != EINTR) break;
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) break;
        struct timespec ts = {0, (++polls < 200) ? 20000L : 200000L};
        (void)nanosleep(&ts, nullptr);
    }
    const long long waited_us =
        (long long)std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    if (got) {
        char buf[32];
        const int n = snprintf(buf, sizeof buf, "%ld\n", (long)getpid());
        if (pwrite(fd, buf, (size_t)n, 0) == n) (void)ftruncate(fd, n);      // who holds it (diagnostics of a time-out elsewhere)
        g_share.lock_acquisitions += 1;
        g_share.lock_wait_us += waited_us;
    }
    lk.lock();
    L.acquiring = false;
    L.held = got;
    if (!got) --L.refs;
    L.cv.notify_all();
    if (!got) {
        if (took) *took = false;
        char who[32] = {0};
        const ssize_t n = pread(fd, who, sizeof who - 1, 0);
        for (ssize_t i = 0; i < n; ++i)
            if (who[i] == '\n') who[i] = 0;
        g_err = "psoap: the device lock " + L.path + " was not released within " + std::to_string((int)limit) +
                " s (PSOAP_DEVICE_LOCK_TIMEOUT_S); last holder: pid " + (n > 0 ? who : "unknown");
        return 2;
    }
    return 0;
}

static void device_lock_release(int device)
{
    if (!device_lock_enabled()) return;
    DeviceLock* Lp = nullptr;
    {
        std::lock_guard<std::mutex> g(g_devlock_mu);
        auto it = g_devlocks.find(device);
        if (it == g_devlocks.end()) return;
        Lp = it->second;
    }
    std::lock_guard<std::mutex> lk(Lp->mu);
    if (Lp->fd < 0 || Lp->pid != getpid() || Lp->refs <= 0) return;
    if (--Lp->refs == 0 && Lp->held) {
        (void)flock(Lp->fd, LOCK_UN);
        Lp->held = false;
    }
}

// How many cooperating processes use a device: each holds one of 64 slot files (<dir>/gpu_<bus>.slot<k>, locked for the
// life of the process; PSOAP_DEVICE_SLOTS=0: no count).  Counted whether or not the lock is on: without the lock the count
// is what sends evaluations down the staged path.
struct SlotState {
    int fd = -1;
    pid_t pid = 0;
    int procs = 1;
    std::chrono::steady_clock::time_point counted{};
    bool warned = false;
};
static std::map<int, SlotState> g_slots;
static bool device_slots_enabled()
{
    static const bool on = !(getenv("PSOAP_DEVICE_SLOTS") && getenv("PSOAP_DEVICE_SLOTS")[0] == '0');
    return on;
}
static void device_slot_take(int device)
{
    if (!device_slots_enabled() || share_dir().empty()) return;
    std::lock_guard<std::mutex> g(g_devlock_mu);
    SlotState& S = g_slots[device];
    if (S.pid == getpid() && S.fd >= 0) return;
    if (S.fd >= 0) (void)close(S.fd);       // (inherited through fork(): shares its lock with the parent)
    S = SlotState();
    S.pid = getpid();
    for (int k = 0; k < 64; ++k) {
        const int fd = open_share_file(device_file(device, (".slot" + std::to_string(k)).c_str()), true);
        if (fd < 0) break;
        if (flock(fd, LOCK_EX | LOCK_NB) == 0) {
            S.fd = fd;
            break;
        }
        (void)close(fd);
    }
}
// The driver's own account of who uses the device: every process that has opened /dev/kfd appears under
// /sys/class/kfd/kfd/proc/<pid>/ with one entry per hardware queue (queues/<id>/gpuid).  Counting the processes with a queue
// on THIS device sees what the slot files cannot: programs that do not go through this library (a torch job, another
// user's work) -- they, too, take one of the eight process contexts the device keeps mapped.  -1 where sysfs does not say
// (no KFD, a container without it, PSOAP_KFD_COUNT=0).  The device's KFD id comes from the topology node whose PCI location
// matches hipDeviceGetPCIBusId.
static int kfd_procs_on_device(int device)
{
    static const bool off = getenv("PSOAP_KFD_COUNT") && getenv("PSOAP_KFD_COUNT")[0] == '0';
    if (off) return -1;
    static std::map<int, std::string> gpu_ids;          // per HIP device: its KFD gpu_id ("" = unknown); guarded by g_devlock_mu
    auto it = gpu_ids.find(device);
    if (it == gpu_ids.end()) {
        std::string found;
        char bus[64] = {0};
        unsigned int dom = 0, b = 0, d = 0, f = 0;
        if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, device) == hipSuccess && sscanf(bus, "%x:%x:%x.%x", &dom, &b, &d, &f) == 4) {
            const unsigned long want_loc = ((unsigned long)b << 8) | ((unsigned long)d << 3) | (unsigned long)f;
            if (DIR* nodes = opendir("/sys/class/kfd/kfd/topology/nodes")) {
                while (struct dirent* e = readdir(nodes)) {
                    if (e->d_name[0] == '.') continue;
                    const std::string nd = std::string("/sys/class/kfd/kfd/topology/nodes/") + e->d_name;
                    FILE* fp = fopen((nd + "/properties").c_str(), "r");
                    if (!fp) continue;
                    char key[64];
                    unsigned long long val = 0, loc = ~0ull, domain = 0, simd = 0;
                    while (fscanf(fp, "%63s %llu", key, &val) == 2) {
                        if (!strcmp(key, "location_id")) loc = val;
                        else if (!strcmp(key, "domain")) domain = val;
                        else if (!strcmp(key, "simd_count")) simd = val;
                    }
                    fclose(fp);
                    if (simd == 0 || loc != want_loc || domain != dom) continue;
                    if (FILE* fg = fopen((nd + "/gpu_id").c_str(), "r")) {
                        char id[32] = {0};
                        if (fscanf(fg, "%31s", id) == 1) found = id;
                        fclose(fg);
                    }
                }
                closedir(nodes);
            }
        }
        it = gpu_ids.emplace(device, found).first;
    }
    if (it->second.empty()) return -1;
    DIR* procs = opendir("/sys/class/kfd/kfd/proc");
    if (!procs) return -1;
    int n = 0;
    while (struct dirent* e = readdir(procs)) {
        if (e->d_name[0] < '0' || e->d_name[0] > '9') continue;
        const std::string qd = std::string("/sys/class/kfd/kfd/proc/") + e->d_name + "/queues";
        DIR* qs = opendir(qd.c_str());
        if (!qs) continue;
        bool here = false;
        while (struct dirent* q = readdir(qs)) {
            if (q->d_name[0] == '.' || here) continue;
            if (FILE* fg = fopen((qd + "/" + q->d_name + "/gpuid").c_str(), "r")) {
                char id[32] = {0};
                if (fscanf(fg, "%31s", id) == 1 && it->second == id) here = true;
                fclose(fg);
            }
        }
        closedir(qs);
        n += here ? 1 : 0;
    }
    closedir(procs);
    return n;
}

// processes on this device: those with a slot file (this library's), or -- where the driver says -- all that hold a hardware
// queue on it (kfd_procs_on_device), whichever is more; this one included; re-counted at most four times a second
static int share_procs(int device)
{
    if (const char* e = getenv("PSOAP_SHARE_PROCS"))      // tests: pretend
        if (atoi(e) > 0) return atoi(e);
    if (!device_slots_enabled() || share_dir().empty()) return 1;
    std::lock_guard<std::mutex> g(g_devlock_mu);
    SlotState& S = g_slots[device];
    if (S.pid != getpid() || S.fd < 0) return 1;
    const auto now = std::chrono::steady_clock::now();
    if (S.counted.time_since_epoch().count() != 0 && std::chrono::duration<double>(now - S.counted).count() < 0.25) return S.procs;
    int n = 0;
    for (int k = 0; k < 64; ++k) {
        const int fd = open_share_file(device_file(device, (".slot" + std::to_string(k)).c_str()), false);
        if (fd < 0) break;                              // slot files are created in order and never removed
        if (flock(fd, LOCK_EX | LOCK_NB) == 0) (void)flock(fd, LOCK_UN);
        else ++n;                                       // held: by another process, or by this one's own descriptor
        (void)close(fd);
    }
    const int k = kfd_procs_on_device(device);
    if (k > n) n = k;
    S.procs = n > 0 ? n : 1;
    S.counted = now;
    return S.procs;
}

// Which path an evaluation takes on a device that `procs` processes share.
static int share_dag_max()
{
    const char* e = getenv("PSOAP_SHARE_DAG_MAX");
    return (e && atoi(e) > 0) ? atoi(e) : 8;
}
static int share_retries()
{
    const char* e = getenv("PSOAP_SHARE_RETRIES");
    return (e && atoi(e) >= 0) ? atoi(e) : 3;
}
// experiments (tools/shared_gpu_probe.py): disturbed launches are counted but their values handed out all the same -- is
// every wrong value one of a launch that reported a moved workgroup?
static bool share_detect_only()
{
    static const bool on = getenv("PSOAP_SHARE_DETECT_ONLY") && getenv("PSOAP_SHARE_DETECT_ONLY")[0] == '1';
    return on;
}
// The decision is taken ONCE per entry into the library and kept for the whole call (DeviceScope below): the process count
// is re-read four times a second, and a call that asked twice -- once to decide whether it needs the device lock, once to
// choose the path -- could get two answers: a persistent launch issued WITHOUT the lock (review of round 5).
static thread_local int g_share_decision_depth = 0;       // > 0: inside an entry point
static thread_local std::map<int, bool>* g_share_decision = nullptr;
static bool share_wants_staged_now(int device);
static bool share_wants_staged(int device)
{
    if (g_share_decision_depth > 0 && g_share_decision) {
        auto it = g_share_decision->find(device);
        if (it != g_share_decision->end()) return it->second;
        const bool v = share_wants_staged_now(device);
        (*g_share_decision)[device] = v;
        return v;
    }
    return share_wants_staged_now(device);
}
static void share_warn_unsafe_regime(int procs, bool pinned_dag);
static bool share_wants_staged_now(int device)
{
    static const int pinned = [] {
        const char* e = getenv("PSOAP_SHARE_POLICY");
        return !e ? 0 : (!strcmp(e, "dag") ? 1 : (!strcmp(e, "staged") ? 2 : 0));
    }();
    if (pinned) {
        if (pinned == 1) share_warn_unsafe_regime(share_procs(device), true);
        return pinned == 2;
    }
    const int procs = share_procs(device);
    if (procs <= 1) return false;
    if (!device_lock_enabled()) return true;            // nobody keeps two persistent launches apart
    if (procs <= share_dag_max()) {
        share_warn_unsafe_regime(procs, false);
        return false;
    }
    static std::atomic<bool> hinted{false};
    if (!hinted.exchange(true) && !(getenv("PSOAP_QUIET") && getenv("PSOAP_QUIET")[0] == '1'))
        fprintf(stderr,
                "psoap: %d processes share this GPU: evaluations take the staged path (safe, slower).  PSOAP_GPU_SERVER=auto lets "
                "ONE process own the device and evaluate all workers' calls in group launches (psoap_amd/server.py: 4-9 x the "
                "rate beyond 8 workers).\n", procs);
    return true;
}
// tests: every k-th launch is treated as tainted (PSOAP_TEST_TAINT_EVERY=k), to drive the retry / fallback logic on a
// device nobody shares
static bool share_inject_taint()
{
    static const int every = getenv("PSOAP_TEST_TAINT_EVERY") ? atoi(getenv("PSOAP_TEST_TAINT_EVERY")) : 0;
    static std::atomic<long long> n{0};
    return every > 0 && (++n % every) == 0;
}

// Persistent launches among MORE than 8 process contexts are outside what was measured clean: a user who pins
// them there (PSOAP_SHARE_POLICY=dag, PSOAP_SHARE_DAG_MAX > 8) or asks for tainted values (PSOAP_SHARE_DETECT_ONLY=1) is told
// so, once.
static void share_warn_unsafe_regime(int procs, bool pinned_dag)
{
    static std::atomic<bool> warned{false};
    const bool beyond = procs > 8 && (pinned_dag || share_dag_max() > 8);
    if (!(beyond || share_detect_only()) || warned.exchange(true)) return;
    if (getenv("PSOAP_QUIET") && getenv("PSOAP_QUIET")[0] == '1') return;
    if (share_detect_only())
        fprintf(stderr, "psoap: PSOAP_SHARE_DETECT_ONLY=1: evaluations that reported a moved workgroup are handed out as they are "
                        "(an experiment's setting: such values may be wrong).\n");
    if (beyond)
        fprintf(stderr, "psoap: %d processes share this GPU and the persistent kernel was pinned there (PSOAP_SHARE_POLICY=dag or "
                        "PSOAP_SHARE_DAG_MAX > 8): the moved-workgroup check was measured clean only up to 8 processes; beyond, "
                        "the staged path or PSOAP_GPU_SERVER=auto is the supported route.\n", procs);
}

struct DeviceScope {
    int dev;
    bool ok, took = false;
    std::map<int, bool> decisions;          // share_wants_staged per device, fixed for the duration of this call
    bool outermost = false;
    explicit DeviceScope(int d) : dev(d)
    {
        if (g_share_decision_depth++ == 0) {
            g_share_decision = &decisions;
            outermost = true;
        }
        ok = device_lock_acquire(d, &took) == 0;
    }
    ~DeviceScope()
    {
        if (took) device_lock_release(dev);
        if (--g_share_decision_depth == 0 && outermost) g_share_decision = nullptr;
    }
    DeviceScope(const DeviceScope&) = delete;
    DeviceScope& operator=(const DeviceScope&) = delete;
};
// every entry point that touches the device: the lock for the duration of the call, an error when it cannot be had
#define DEVICE_SCOPE(d)      \
    DeviceScope scope_(d);   \
    if (!scope_.ok) return 2
// the destroy entry points: the resources go whether or not the lock could be had (a time-out there must not leak device
// memory, nor make the caller's close() raise: hipDeviceSynchronize + hipFree disturb nobody's persistent launch)
#define DEVICE_SCOPE_DESTROY(d) DeviceScope scope_(d)

// Owning device / pinned-host pointer: early returns free whatever was allocated so far.
template <class T>
struct DevBuf {
    T* p = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t count) { return hipMalloc(reinterpret_cast<void**>(&p), sizeof(T) * (count ? count : 1)); }
    operator T*() const { return p; }
};

// One uploaded batch of proposals.  A handle keeps two, so the proposals of step k+1 travel over
// PCIe (copy stream) while the persistent kernel still factors step k: psoap_batch_upload fills the slot
// that is not being evaluated, psoap_batch_eval promotes the pending slot.
struct BatchSlot {
    double* dLwl = nullptr;    // max_batch x 3 x N
    double* dGp = nullptr;     // max_batch x 6
    int* dTooFast = nullptr;   // max_batch: |v| >= c flags (orbit proposals)
    bool toofast_dirty = false;  // an orbit upload may have raised flags: clear before the slot is reused
    DagMat* dMats = nullptr;   // per-matrix records of this slot (max_batch entries)
    int mats_B = 0, mats_C = 0;
    int B = 0, C = 0;
    double mu = 1.0;
    std::vector<char> neg;     // per-proposal: a hyper-parameter was negative -> -inf
    hipEvent_t evUpload = nullptr;   // copy stream: H2D (+ Doppler shift) of this slot complete
    hipEvent_t evEval = nullptr;     // compute stream: last evaluation that read this slot complete
};

namespace psoap { struct PredictWs; }

// Streamed evaluation (dag_kernel.hpp, "Streamed evaluation"): one resident launch of the persistent kernel, matrices
// come and go through `lanes` workspaces of the handle.  Host side: lane allocation, the submission ring in pinned
// memory, launch / relaunch, result polling.
struct StreamState {
    bool open = false;
    int C = 0, lanes = 0, scheme = 0;
    size_t h_stride = 0;               // doubles per lane in hLw
    unsigned int n_tasks = 0, ctrs_per_lane = 0, slots_per_lane = 0;
    DagQueues queues{};
    DagTask* dTasks = nullptr;
    StreamLane* dLanes = nullptr;
    StreamDev* dDev = nullptr;
    StreamHost* hHost = nullptr;       // pinned, host-coherent
    double* hLw = nullptr;             // pinned proposals, lane-major
    double* hGp = nullptr;
    double* dWs = nullptr;             // lanes x slots_per_lane partial tiles
    char* dDag = nullptr;              // DagCtl, MatFlags[lanes], arrival counters[lanes x ctrs_per_lane]
    size_t arrive_off = 0;
    DagMat* dMats = nullptr;
    unsigned long long* dTlog = nullptr;
    unsigned int tlog_cap = 0;
    hipEvent_t evStart = nullptr, evExit = nullptr;   // around the resident launch (timed: the roofline of bench.py)
    bool launched = false;
    unsigned long long completed_before = 0;     // StreamDev::completed when the current launch started
    double last_launch_ms = 0.0;                 // the launch that ended last (psoap_stream_pause / close measure it)
    long long last_launch_matrices = 0;
    unsigned long long head = 0;                 // submissions published
    std::vector<long long> lane_ticket;          // ticket held by each lane (what the caller knows it by), -1: free
    std::vector<unsigned long long> lane_seq;    // submission number the lane's matrix runs under NOW: the ticket, until a
                                                 // tainted result made the library submit the lane again (stream_lane_result)
    std::vector<int> lane_tries;                 // resubmissions of the lane's current ticket
    std::vector<int> lane_kind;                  // StreamEntry::kind and ::mu of the lane's submission (for a resubmission)
    std::vector<double> lane_mu;
    std::vector<char> neg;                       // per ring entry: a hyper-parameter was negative -> -inf
    long long launches = 0;
    double idle_ms = 20.0;
};
// (Round 5 measured a resident launch that leaves workgroup slots FREE for kernels of other streams -- the RCCL all_gather
// of the walker lnprobs and its staging copies: 0, 4 or 8 of the 512 slots unoccupied, the gather per half-ensemble still
// waits for the launch to leave, 90.6-93.2 ms per step against 37.9 without a collective (profiles/r5_gather_beside_stream.txt).
// Device collectives and a resident launch do not mix: several ranks use the launch-per-step path, or gather on the host.)

struct psoap_group;
struct psoap_chunk {
    int device = 0;
    int N = 0, Npad = 0, ld = 0, P = 0;
    int max_batch = 0;
    size_t mat_stride = 0;
    // resident data
    double* dFl = nullptr;
    double* dSigma = nullptr;
    double* dGrid = nullptr;
    int32_t* dEpoch = nullptr;
    int n_epochs = 0;
    // workspaces
    double* dK = nullptr;    // max_batch x Npad x ld
    double* dWt = nullptr;   // max_batch x 2 x 128 x 128 (the persistent kernel alternates two per matrix)
    double* dR = nullptr;    // max_batch x Npad
    MatAcc* dAcc = nullptr;  // max_batch
    double* dVel = nullptr;  // max_batch x 3 x n_epochs
    double* dOut = nullptr;  // max_batch
    double* dDates = nullptr;  // n_epochs observation dates (orbit proposals)
    double* dPorb = nullptr;   // max_batch x 13 orbital parameters
    double* hPorb = nullptr;
    char* dDag = nullptr;    // DagCtl followed by max_batch MatFlags (zeroed before every DAG launch)
    unsigned int* hDagErr = nullptr;
    int mode = 1;            // 1 = persistent DAG kernel, 0 = staged panels
    int dag_grid = 0, n_cus = 0;   // persistent workgroups the device admits; compute units
    int plan_workers = 0;          // workgroups of the current task list (dag_pick_workers)
    // task list of the persistent kernel for the current batch size (dag_build_tasks)
    int plan_B = 0, plan_scheme = 0;
    unsigned int plan_tasks = 0, plan_ctrs = 0, plan_slots = 0;
    DagQueues plan_queues{};
    DagTask* dTasks = nullptr;
    unsigned int* dOrder = nullptr;   // ready-only hand-out (DagPool): order[] and dep[], tasks_cap entries each
    unsigned int* dDep = nullptr;
    unsigned int plan_n_main[DAG_QUEUES] = {};
    bool plan_pool = false;
    size_t tasks_cap = 0;
    double* dWs = nullptr;   // split-K partial tiles, plan_slots x 128 x 128
    size_t ws_cap = 0;
    size_t arrive_off = 0;   // byte offset of the arrival counters inside dDag
    size_t arrive_cap = 0;   // ints
    unsigned long long* dTlog = nullptr;  // optional per-task timestamps (debug)
    long long tlog_tasks = 0;
    // pinned host staging (one set: reused once the previous upload's copies have completed)
    double* hLwl = nullptr;
    double* hGp = nullptr;
    double* hVel = nullptr;
    double* hOut = nullptr;
    // proposal batches
    BatchSlot slot[2];
    int act = -1;            // slot of the last / running evaluation
    int pend = -1;           // uploaded, not yet evaluated
    hipStream_t copy = nullptr;
    hipEvent_t evStaging = nullptr;   // copy stream: the pinned staging buffers have been consumed
    hipEvent_t evLast = nullptr;      // the most recent evaluation of this handle, whichever slot and stream it used:
    bool last_recorded = false;       // the shared workspaces (K, r, Wt, acc, out) are free once it has completed
    // execution
    int groups = 2;
    hipStream_t streams[MAX_GROUPS] = {};
    hipEvent_t evDone[MAX_GROUPS] = {};
    // profiling
    bool profiling = false;
    std::vector<hipEvent_t> evPool;
    struct Rec {
        int cls;
        int e0, e1;
        double flops, bytes;
    };
    std::vector<Rec> recs;
    psoap_timings last = {};
    // predict workspace (grow-only; psoap_chunk_predict)
    psoap::PredictWs* pws = nullptr;
    // streamed evaluation (psoap_stream_*)
    StreamState stream;
    bool dev_locked = false;     // this handle holds a reference on the device's inter-process lock (device_lock_acquire)
    int last_path = 1;           // what the evaluation in flight runs on: 1 the persistent kernel, 0 the staged path
    struct psoap_group* last_group = nullptr;   // the group launch that evaluation belongs to (nullptr: the handle's own)
};

static int handle_lock(psoap_chunk* h)
{
    if (!h->dev_locked) {
        bool took = false;
        if (int rc = device_lock_acquire(h->device, &took)) return rc;
        h->dev_locked = took;
    }
    return 0;
}
static void handle_unlock(psoap_chunk* h, bool force = false)
{
    if (!force)
        for (long long t : h->stream.lane_ticket)
            if (t >= 0) return;           // a stream with tickets outstanding keeps the device
    if (h->dev_locked) {
        device_lock_release(h->device);
        h->dev_locked = false;
    }
}

static int set_dev(const psoap_chunk* h) { HIP_TRY(hipSetDevice(h->device)); return 0; }

extern "C" int psoap_version(void) { return 1; }
extern "C" const char* psoap_last_error(void) { return g_err.c_str(); }

extern "C" int psoap_device_count(int* count)
{
    HIP_TRY(hipGetDeviceCount(count));
    return 0;
}

// What sharing the device with other processes has cost this process so far (include/psoap_gp.h: PSOAP_SHARE_*).
extern "C" int psoap_share_stats(int device, long long* out, int n)
{
    if (!out || n < 1) FAIL("psoap_share_stats: bad arguments");
    const long long v[PSOAP_SHARE_N] = {
        (long long)share_procs(device), g_share.dag_launches.load(), g_share.tainted.load(), g_share.retries.load(),
        g_share.staged_fallbacks.load(), g_share.staged_policy.load(), g_share.moved_tasks.load(), g_share.moved_xcd.load(),
        g_share.lock_acquisitions.load(), g_share.lock_wait_us.load(), g_share.stream_resubmits.load(),
        (long long)(device_lock_enabled() ? 1 : 0)};
    for (int k = 0; k < n && k < PSOAP_SHARE_N; ++k) out[k] = v[k];
    return 0;
}

// hipFuncSetAttribute applies to the CURRENT device's function object: one pass per device, under a lock
// (a process may open handles on several GPUs, from several host threads).
static int configure_kernels(int device)
{
    static std::mutex mu;
    static std::vector<char> done;
    std::lock_guard<std::mutex> lock(mu);
    if (device < 0) FAIL("negative device index");
    if ((size_t)device < done.size() && done[device]) return 0;
    const int lds = (int)GEMM_LDS_BYTES;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_panel_update),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_trsm_strip),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds));
#define PSOAP_SET_LDS(...) \
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(__VA_ARGS__), hipFuncAttributeMaxDynamicSharedMemorySize, lds))
    PSOAP_SET_LDS(k_chol_dag<1, false, false, false, DAG_WPE_TP>);
    PSOAP_SET_LDS(k_chol_dag<2, false, false, false, DAG_WPE_TP>);
    PSOAP_SET_LDS(k_chol_dag<3, false, false, false, DAG_WPE_TP>);
    PSOAP_SET_LDS(k_chol_dag<1, false, true>);
    PSOAP_SET_LDS(k_chol_dag<2, false, true>);
    PSOAP_SET_LDS(k_chol_dag<3, false, true>);
    PSOAP_SET_LDS(k_chol_dag<1, false, true, false, 1>);
    PSOAP_SET_LDS(k_chol_dag<2, false, true, false, 1>);
    PSOAP_SET_LDS(k_chol_dag<3, false, true, false, 1>);
    PSOAP_SET_LDS(k_chol_solo<1>);
    PSOAP_SET_LDS(k_chol_solo<2>);
    PSOAP_SET_LDS(k_chol_solo<3>);
    PSOAP_SET_LDS(k_chol_dag<1, false, false, true>);
    PSOAP_SET_LDS(k_chol_dag<2, false, false, true>);
    PSOAP_SET_LDS(k_chol_dag<3, false, false, true>);
    PSOAP_SET_LDS(k_chol_dag<1, false, true, true>);
    PSOAP_SET_LDS(k_chol_dag<2, false, true, true>);
    PSOAP_SET_LDS(k_chol_dag<3, false, true, true>);
#undef PSOAP_SET_LDS
    HIP_TRY(predict_configure_kernels());
    if (done.size() <= (size_t)device) done.resize((size_t)device + 1, 0);
    done[device] = 1;
    return 0;
}

// hipSetDevice + the per-device kernel attributes: the first thing every entry point that launches does
static int enter_device(int device)
{
    HIP_TRY(hipSetDevice(device));
    return configure_kernels(device);
}

// persistent workgroups of the dependency-graph kernel on this device (2 per CU when they fit)
static int dag_workers(int device, int* out, int* cus = nullptr)
{
    int blocks_per_cu = 0;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, k_chol_dag<3, true, true>, GEMM_THREADS,
                                                         GEMM_LDS_BYTES));
    if (blocks_per_cu < 1) blocks_per_cu = 1;
    if (blocks_per_cu > 2) blocks_per_cu = 2;
    if (DAG_WPE_TP > 2) blocks_per_cu = DAG_WPE_TP;      // (-DPSOAP_WPE3: the throughput kernels are compiled for three)
    *out = blocks_per_cu * prop.multiProcessorCount;
    if (cus) *cus = prop.multiProcessorCount;
    return 0;
}

static int chunk_alloc(psoap_chunk* h, const double* fl, const double* sigma)
{
    const int N = h->N;
    const size_t nb = (size_t)h->max_batch;
    HIP_TRY(hipMalloc(&h->dFl, sizeof(double) * N));
    HIP_TRY(hipMalloc(&h->dSigma, sizeof(double) * N));
    HIP_TRY(hipMalloc(&h->dK, sizeof(double) * nb * h->mat_stride));
    HIP_TRY(hipMalloc(&h->dWt, sizeof(double) * nb * WT_STRIDE));       // two Wt tiles + the mailbox per matrix
    HIP_TRY(hipMemset(h->dWt, 0, sizeof(double) * nb * WT_STRIDE));   // the strictly upper part of every W stays zero
    HIP_TRY(hipMalloc(&h->dR, sizeof(double) * nb * h->Npad));
    HIP_TRY(hipMalloc(&h->dAcc, sizeof(MatAcc) * ACC_ROWS * nb));      // per matrix: one record per block row (common.hpp)
    HIP_TRY(hipMalloc(&h->dOut, sizeof(double) * nb));
    HIP_TRY(hipMalloc(&h->dPorb, sizeof(double) * nb * 13));
    HIP_TRY(hipHostMalloc(&h->hPorb, sizeof(double) * nb * 13));
    HIP_TRY(hipHostMalloc(&h->hLwl, sizeof(double) * nb * 3 * N));
    HIP_TRY(hipHostMalloc(&h->hGp, sizeof(double) * nb * 6));
    HIP_TRY(hipHostMalloc(&h->hOut, sizeof(double) * nb));
    for (BatchSlot& sl : h->slot) {
        HIP_TRY(hipMalloc(&sl.dLwl, sizeof(double) * nb * 3 * N));
        HIP_TRY(hipMalloc(&sl.dGp, sizeof(double) * nb * 6));
        HIP_TRY(hipMalloc(&sl.dTooFast, sizeof(int) * nb));
        HIP_TRY(hipMemset(sl.dTooFast, 0, sizeof(int) * nb));
        HIP_TRY(hipMalloc(&sl.dMats, sizeof(DagMat) * nb));
        HIP_TRY(hipEventCreateWithFlags(&sl.evUpload, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&sl.evEval, hipEventDisableTiming));
    }
    h->arrive_off = sizeof(DagCtl) + sizeof(MatFlags) * nb;
    // (arrival counters, and behind them the `taken` bitmap of the ready-only hand-out: one bit per task, at most 9 parts
    // per tile + the early diagonal parts)
    h->arrive_cap = nb * (size_t)h->P * (h->P + 1) / 2 + 16 + (9 * nb * (size_t)h->P * (h->P + 1) / 2 + 1024) / 32 + 8;
    HIP_TRY(hipMalloc(&h->dDag, h->arrive_off + sizeof(int) * h->arrive_cap));
    HIP_TRY(hipHostMalloc(&h->hDagErr, 64));
    h->hDagErr[0] = 0;
    if (int rc = dag_workers(h->device, &h->dag_grid, &h->n_cus)) return rc;
    // one compute stream per handle; the extra streams of the staged mode's groups are created on first
    // use, so that the streams of several handles spread over the runtime's hardware queues (handles that
    // evaluate concurrently must not share one).  Uploads run on a stream of their own.
    HIP_TRY(hipStreamCreateWithFlags(&h->streams[0], hipStreamNonBlocking));
    // The uploads share the evaluation's stream.  (Rounds 1-3 gave them a stream of their own; the copies of the next
    // proposals are blit kernels that get compute units only when the persistent launch leaves, so nothing overlapped
    // anyway -- and every stream is a hardware queue: with three per process, eight worker processes on one GPU
    // oversubscribe the device's queue slots, and the scheduler then rotates its run list under running kernels: 5.6-10 ms
    // per single evaluation instead of 2.6, and a wrong value in 1-2 of 36,000 even with the device lock.  With two queues
    // per process: 2.64 ms, none in 36,000.  A handle that uploads WHILE it evaluates -- the pipelined loops of bench.py and
    // EnsembleEvaluator, one process per GPU -- gets its copy stream then (upload_begin: 0.6 % on a 32-walker step);
    // PSOAP_COPY_STREAM=1 creates it here, =0 never.)
    if (getenv("PSOAP_COPY_STREAM") && getenv("PSOAP_COPY_STREAM")[0] == '1')
        HIP_TRY(hipStreamCreateWithFlags(&h->copy, hipStreamNonBlocking));
    else
        h->copy = h->streams[0];
    for (int g = 0; g < MAX_GROUPS; ++g) HIP_TRY(hipEventCreateWithFlags(&h->evDone[g], hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&h->evStaging, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&h->evLast, hipEventDisableTiming));
    HIP_TRY(hipMemcpy(h->dFl, fl, sizeof(double) * N, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->dSigma, sigma, sizeof(double) * N, hipMemcpyHostToDevice));
    return 0;
}

extern "C" int psoap_chunk_create(psoap_chunk** out, int device, int N, const double* fl, const double* sigma,
                                  int max_batch)
{
    if (!out || N <= 0 || max_batch <= 0 || !fl || !sigma) FAIL("psoap_chunk_create: bad arguments");
    *out = nullptr;
    DEVICE_SCOPE(device);
    if (int rc = enter_device(device)) return rc;
    device_slot_take(device);
    psoap_chunk* h = new psoap_chunk();
    h->device = device;
    h->N = N;
    h->Npad = round_up(N, NB);
    h->ld = h->Npad;
    h->P = h->Npad / NB;
    h->max_batch = max_batch;
    h->mat_stride = (size_t)h->Npad * h->ld;
    if (int rc = chunk_alloc(h, fl, sigma)) {
        // e.g. out of memory half way: give everything back (the message of the failing call is kept)
        const std::string keep = g_err;
        (void)psoap_chunk_destroy(h);
        g_err = keep;
        return rc;
    }
    *out = h;
    return 0;
}



extern "C" int psoap_stream_close(psoap_chunk* h);

extern "C" int psoap_chunk_destroy(psoap_chunk* h)
{
    if (!h) return 0;
    DEVICE_SCOPE_DESTROY(h->device);
    (void)hipSetDevice(h->device);
    if (h->stream.open) (void)psoap_stream_close(h);
    (void)hipDeviceSynchronize();
    handle_unlock(h, true);
    (void)hipFree(h->dFl); (void)hipFree(h->dSigma); (void)hipFree(h->dGrid); (void)hipFree(h->dEpoch);
    (void)hipFree(h->dK); (void)hipFree(h->dWt); (void)hipFree(h->dR); (void)hipFree(h->dAcc);
    (void)hipFree(h->dVel); (void)hipFree(h->dOut);
    (void)hipFree(h->dDag); (void)hipHostFree(h->hDagErr); (void)hipFree(h->dTlog);
    (void)hipFree(h->dTasks); (void)hipFree(h->dWs); (void)hipFree(h->dOrder); (void)hipFree(h->dDep);
    (void)hipFree(h->dDates); (void)hipFree(h->dPorb); (void)hipHostFree(h->hPorb);
    (void)hipHostFree(h->hLwl); (void)hipHostFree(h->hGp); (void)hipHostFree(h->hVel); (void)hipHostFree(h->hOut);
    for (BatchSlot& sl : h->slot) {
        (void)hipFree(sl.dLwl); (void)hipFree(sl.dGp); (void)hipFree(sl.dTooFast); (void)hipFree(sl.dMats);
        if (sl.evUpload) (void)hipEventDestroy(sl.evUpload);
        if (sl.evEval) (void)hipEventDestroy(sl.evEval);
    }
    for (int g = 0; g < MAX_GROUPS; ++g) {
        if (h->streams[g]) (void)hipStreamDestroy(h->streams[g]);
        if (h->evDone[g]) (void)hipEventDestroy(h->evDone[g]);
    }
    if (h->copy && h->copy != h->streams[0]) (void)hipStreamDestroy(h->copy);
    if (h->evStaging) (void)hipEventDestroy(h->evStaging);
    if (h->evLast) (void)hipEventDestroy(h->evLast);
    for (auto e : h->evPool) (void)hipEventDestroy(e);
    delete h->pws;
    delete h;
    return 0;
}

extern "C" int psoap_chunk_set_data(psoap_chunk* h, const double* fl, const double* sigma)
{
    if (!h || !fl || !sigma) FAIL("psoap_chunk_set_data: bad arguments");
    if (h->stream.open) FAIL("psoap_chunk_set_data: the handle has an open stream (psoap_stream_close first)");
    DEVICE_SCOPE(h->device);
    if (set_dev(h)) return 1;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(h->dFl, fl, sizeof(double) * h->N, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->dSigma, sigma, sizeof(double) * h->N, hipMemcpyHostToDevice));
    return 0;
}

extern "C" int psoap_chunk_set_grid(psoap_chunk* h, const double* lwl, const int32_t* epoch, int n_epochs)
{
    if (!h || !lwl || !epoch || n_epochs <= 0) FAIL("psoap_chunk_set_grid: bad arguments");
    if (h->stream.open) FAIL("psoap_chunk_set_grid: the handle has an open stream (psoap_stream_close first)");
    for (int i = 0; i < h->N; ++i)
        if (epoch[i] < 0 || epoch[i] >= n_epochs) FAIL("psoap_chunk_set_grid: epoch index out of range");
    DEVICE_SCOPE(h->device);
    if (set_dev(h)) return 1;
    HIP_TRY(hipDeviceSynchronize());
    if (!h->dGrid) HIP_TRY(hipMalloc(&h->dGrid, sizeof(double) * h->N));
    if (!h->dEpoch) HIP_TRY(hipMalloc(&h->dEpoch, sizeof(int32_t) * h->N));
    if (h->dVel) { HIP_TRY(hipFree(h->dVel)); h->dVel = nullptr; }
    if (h->hVel) { HIP_TRY(hipHostFree(h->hVel)); h->hVel = nullptr; }
    HIP_TRY(hipMalloc(&h->dVel, sizeof(double) * (size_t)h->max_batch * 3 * n_epochs));
    HIP_TRY(hipHostMalloc(&h->hVel, sizeof(double) * (size_t)h->max_batch * 3 * n_epochs));
    h->n_epochs = n_epochs;
    HIP_TRY(hipMemcpy(h->dGrid, lwl, sizeof(double) * h->N, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->dEpoch, epoch, sizeof(int32_t) * h->N, hipMemcpyHostToDevice));
    return 0;
}

extern "C" int psoap_chunk_set_stream_groups(psoap_chunk* h, int groups)
{
    if (!h || groups < 1 || groups > MAX_GROUPS) FAIL("psoap_chunk_set_stream_groups: 1 <= groups <= 8");
    h->groups = groups;
    return 0;
}

// Debug: allocate a per-task timestamp log for the DAG kernel and read it back
// (4 x 100 MHz stamps per task: start, updated, diagonal ready / factored, end).
extern "C" int psoap_chunk_dag_tasklog(psoap_chunk* h, unsigned long long* out, long long max_tasks)
{
    if (!h) FAIL("null handle");
    DEVICE_SCOPE(h->device);
    if (set_dev(h)) return 1;
    const long long tasks = 9ll * h->max_batch * h->P * (h->P + 1) / 2 + 1024;   // <= 8 parts per tile + early DIAG parts
    if (!h->dTlog) {
        HIP_TRY(hipMalloc(&h->dTlog, sizeof(unsigned long long) * 8 * tasks));
        HIP_TRY(hipMemset(h->dTlog, 0, sizeof(unsigned long long) * 8 * tasks));
        h->tlog_tasks = tasks;
        return 0;
    }
    if (out) {
        const long long n = max_tasks < tasks ? max_tasks : tasks;
        HIP_TRY(hipMemcpy(out, h->dTlog, sizeof(unsigned long long) * 8 * n, hipMemcpyDeviceToHost));
    }
    return 0;
}

// Pure host function (no HIP call): the task list the persistent kernel would run for a batch of
// B matrices with P block rows on `workers` workgroups.  Lets the scheduler be validated on a CPU.
extern "C" int psoap_dag_plan(int B, int P, int workers, void* out, long long max_tasks, long long* n_tasks,
                              long long* n_slots, long long* n_ctrs, unsigned int* queue_first)
{
    if (B < 1 || P < 1 || P > 255 || workers < 1 || !n_tasks) FAIL("psoap_dag_plan: bad arguments");
    DagPlan plan = dag_build_tasks(B, P, workers);
    *n_tasks = (long long)plan.tasks.size();
    if (n_slots) *n_slots = plan.n_slots;
    if (n_ctrs) *n_ctrs = plan.n_ctrs;
    if (queue_first) memcpy(queue_first, plan.queues.first, sizeof plan.queues.first);
    if (out) {
        const long long n = max_tasks < *n_tasks ? max_tasks : *n_tasks;
        memcpy(out, plan.tasks.data(), sizeof(DagTask) * n);
    }
    return 0;
}

// Pure host function: the task list every lane of a stream of `lanes` lanes runs for matrices of P block rows
// (dag_build_lane_plan); DagTask::b carries the burst marks (0x8000: the last ticket of a burst).
extern "C" int psoap_stream_plan(int P, int lanes, int workers, int scheme, void* out, long long max_tasks,
                                 long long* n_tasks, long long* n_slots, long long* n_ctrs, int* scheme_out)
{
    if (P < 1 || P > 255 || lanes < 1 || lanes > STREAM_MAX_LANES || workers < 1 || scheme < -1 || scheme > 2 || !n_tasks)
        FAIL("psoap_stream_plan: bad arguments");
    if (scheme < 0) scheme = dag_auto_scheme(std::vector<int>((size_t)lanes, P));
    DagPlan plan = dag_build_lane_plan(P, lanes, workers, scheme);
    *n_tasks = (long long)plan.tasks.size();
    if (n_slots) *n_slots = plan.n_slots;
    if (n_ctrs) *n_ctrs = plan.n_ctrs;
    if (scheme_out) *scheme_out = plan.scheme;
    if (out) {
        const long long n = max_tasks < *n_tasks ? max_tasks : *n_tasks;
        memcpy(out, plan.tasks.data(), sizeof(DagTask) * n);
    }
    return 0;
}

// One matrix with Mt appended column tiles (predict) and, when Ms > 0, the Ms x Ms tiles of their Schur complement as
// tasks of the same launch (DAG_SCHUR).  scheme: -1 automatic, 0 throughput, 1 latency.
extern "C" int psoap_dag_plan_aug(int P, int Mt, int Ms, int workers, int scheme, void* out, long long max_tasks,
                                  long long* n_tasks, long long* n_slots, long long* n_ctrs, unsigned int* queue_first)
{
    if (P < 1 || Mt < 0 || Ms < 0 || Ms > Mt || P + Mt > 255 || workers < 1 || !n_tasks)
        FAIL("psoap_dag_plan_aug: bad arguments");
    DagPlan plan = dag_build_tasks(1, P, workers, scheme, Mt, Ms);
    *n_tasks = (long long)plan.tasks.size();
    if (n_slots) *n_slots = plan.n_slots;
    if (n_ctrs) *n_ctrs = plan.n_ctrs;
    if (queue_first) memcpy(queue_first, plan.queues.first, sizeof plan.queues.first);
    if (out) {
        const long long n = max_tasks < *n_tasks ? max_tasks : *n_tasks;
        memcpy(out, plan.tasks.data(), sizeof(DagTask) * n);
    }
    return 0;
}

// The same for a heterogeneous batch: matrix b has Ps[b] block rows.
extern "C" int psoap_dag_plan_multi(int B, const int* Ps, int workers, void* out, long long max_tasks,
                                    long long* n_tasks, long long* n_slots, long long* n_ctrs,
                                    unsigned int* queue_first)
{
    if (B < 1 || !Ps || workers < 1 || !n_tasks) FAIL("psoap_dag_plan_multi: bad arguments");
    for (int b = 0; b < B; ++b)
        if (Ps[b] < 1 || Ps[b] > 255) FAIL("psoap_dag_plan_multi: 1 <= P <= 255");
    DagPlan plan = dag_build_tasks(std::vector<int>(Ps, Ps + B), workers);
    *n_tasks = (long long)plan.tasks.size();
    if (n_slots) *n_slots = plan.n_slots;
    if (n_ctrs) *n_ctrs = plan.n_ctrs;
    if (queue_first) memcpy(queue_first, plan.queues.first, sizeof plan.queues.first);
    if (out) {
        const long long n = max_tasks < *n_tasks ? max_tasks : *n_tasks;
        memcpy(out, plan.tasks.data(), sizeof(DagTask) * n);
    }
    return 0;
}

// Pure host function: the task list of a batch (Ps[b] block rows each; Mt appended column tiles and an Ms x Ms Schur block
// for a single matrix: predict) together with the two hand-out orders of the ready-only scheme (DagPool, dag_kernel.hpp):
// order[] (per queue the finals' task indices, then the parts'), dep[] (per final: the position in order[] of the last
// part of its chain, 0xffffffff without one), n_main[8].  Empty orders (*has_pool = 0) for the throughput scheme.
extern "C" int psoap_dag_plan_pool(int B, const int* Ps, int workers, int Mt, int Ms, int scheme, void* tasks_out,
                                   long long max_tasks, long long* n_tasks, unsigned int* order_out, unsigned int* dep_out,
                                   unsigned int* n_main_out, unsigned int* queue_first, int* has_pool, long long* n_ctrs)
{
    if (B < 1 || !Ps || workers < 1 || !n_tasks || Mt < 0 || Ms < 0 || Ms > Mt || (Mt > 0 && B != 1))
        FAIL("psoap_dag_plan_pool: bad arguments");
    for (int b = 0; b < B; ++b)
        if (Ps[b] < 1 || Ps[b] + Mt > 255) FAIL("psoap_dag_plan_pool: 1 <= P (+ Mt) <= 255");
    DagPlan plan = dag_build_tasks(std::vector<int>(Ps, Ps + B), workers, scheme, Mt, Ms);
    if (plan.scheme >= 1 && plan.order.empty()) dag_build_pool(plan);      // (the shipped build does not use the orders)
    *n_tasks = (long long)plan.tasks.size();
    if (has_pool) *has_pool = plan.order.empty() ? 0 : 1;
    if (n_ctrs) *n_ctrs = plan.n_ctrs;
    if (queue_first) memcpy(queue_first, plan.queues.first, sizeof plan.queues.first);
    if (n_main_out) memcpy(n_main_out, plan.n_main, sizeof plan.n_main);
    const long long n = max_tasks < *n_tasks ? max_tasks : *n_tasks;
    if (tasks_out) memcpy(tasks_out, plan.tasks.data(), sizeof(DagTask) * n);
    if (!plan.order.empty()) {
        if (order_out) memcpy(order_out, plan.order.data(), sizeof(unsigned int) * n);
        if (dep_out) memcpy(dep_out, plan.dep.data(), sizeof(unsigned int) * n);
    }
    return 0;
}

// Pure host function: how many persistent workgroups a batch of B matrices (Ps[b] block rows each, Mt appended
// column tiles) gets on a device with `compute_units` CUs that admits `max_workers` of them (dag_pick_workers).
extern "C" int psoap_dag_pick_workers(int B, const int* Ps, int Mt, int compute_units, int max_workers, int* workers)
{
    if (B < 1 || !Ps || Mt < 0 || compute_units < 1 || max_workers < 1 || !workers) FAIL("psoap_dag_pick_workers: bad arguments");
    const std::vector<int> v(Ps, Ps + B);
    int Pmax = 0;
    for (int P : v) Pmax = P > Pmax ? P : Pmax;
    *workers = dag_pick_workers(dag_batch_flops(v, Mt), Pmax, compute_units, max_workers, B);
    return 0;
}

// Debug: copy the current task list (16-byte DagTask records, ticket order) to the host.
extern "C" int psoap_chunk_dag_tasks(psoap_chunk* h, void* out, long long max_tasks, long long* n_tasks)
{
    if (!h || !n_tasks) FAIL("bad arguments");
    DEVICE_SCOPE(h->device);
    if (set_dev(h)) return 1;
    *n_tasks = h->plan_tasks;
    if (out && h->dTasks) {
        const long long n = max_tasks < (long long)h->plan_tasks ? max_tasks : (long long)h->plan_tasks;
        HIP_TRY(hipMemcpy(out, h->dTasks, sizeof(DagTask) * n, hipMemcpyDeviceToHost));
    }
    return 0;
}

extern "C" int psoap_chunk_set_mode(psoap_chunk* h, int mode)
{
    if (!h || mode < 0 || mode > 1) FAIL("psoap_chunk_set_mode: mode must be 0 (staged) or 1 (dag)");
    h->mode = mode;
    return 0;
}

extern "C" int psoap_chunk_set_profiling(psoap_chunk* h, int enabled)
{
    if (!h) FAIL("null handle");
    h->profiling = enabled != 0;
    return 0;
}

// Begin an upload: pick the slot that is not being evaluated, make sure its previous readers and the
// previous user of the pinned staging buffers are done, and record the batch's host-side state.
static int upload_begin(psoap_chunk* h, int B, int c, const double* gp, double mu_GP, BatchSlot** out)
{
    if (h->stream.open) FAIL("the handle has an open stream (psoap_stream_close first): its workspaces belong to the resident launch");
    if (B < 1 || B > h->max_batch) FAIL("batch size outside [1, max_batch]");
    if (c < 1 || c > 3) FAIL("number of components must be 1, 2 or 3");
    const int target = (h->pend >= 0) ? h->pend : (h->act < 0 ? 0 : (h->act ^ 1));
    BatchSlot& sl = h->slot[target];
    // the staging buffers are free once the previous upload's copies have run (they were queued one
    // evaluation ago: normally long finished)
    HIP_TRY(hipEventSynchronize(h->evStaging));
    // an upload under a running evaluation: from now on the copies have a stream of their own (see psoap_chunk_create)
    if (h->copy == h->streams[0] && h->last_recorded && !(getenv("PSOAP_COPY_STREAM") && getenv("PSOAP_COPY_STREAM")[0] == '0')) {
        const hipError_t q = hipEventQuery(h->evLast);
        if (q == hipErrorNotReady) {
            (void)hipGetLastError();
            hipStream_t cs = nullptr;
            HIP_TRY(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
            h->copy = cs;
        } else if (q != hipSuccess) {
            HIP_TRY(q);
        }
    }
    // the slot's device arrays: the evaluation that last read them must be over before they are rewritten
    HIP_TRY(hipStreamWaitEvent(h->copy, sl.evEval, 0));
    sl.B = B;
    sl.C = c;
    sl.mu = mu_GP;
    sl.neg.assign(B, 0);
    for (int b = 0; b < B; ++b)
        for (int k = 0; k < 2 * c; ++k)
            if (gp[(size_t)b * 2 * c + k] < 0.0) sl.neg[b] = 1;  // covariance.py:317,339,362
    memcpy(h->hGp, gp, sizeof(double) * (size_t)B * 2 * c);
    h->pend = target;
    *out = &sl;
    // the copies queued behind this call run beside whatever else is on the device: the handle keeps the device from here
    // to the fetch of the evaluation that reads them (psoap_lnlike: the whole call)
    return handle_lock(h);
}

// The |v| >= c flags of a slot are zero unless an orbit upload used it: only then a memset is queued (it runs
// as a kernel, and a kernel on the copy stream waits for the persistent kernel to leave the device -- the
// plain H2D copies of the proposals do not).
static int clear_too_fast(psoap_chunk* h, BatchSlot& sl)
{
    if (!sl.toofast_dirty) return 0;
    HIP_TRY(hipMemsetAsync(sl.dTooFast, 0, sizeof(int) * (size_t)h->max_batch, h->copy));
    sl.toofast_dirty = false;
    return 0;
}

static int upload_end(psoap_chunk* h, BatchSlot& sl)
{
    HIP_TRY(hipMemcpyAsync(sl.dGp, h->hGp, sizeof(double) * (size_t)sl.B * 2 * sl.C, hipMemcpyHostToDevice, h->copy));
    HIP_TRY(hipEventRecord(h->evStaging, h->copy));
    HIP_TRY(hipEventRecord(sl.evUpload, h->copy));
    return 0;
}

extern "C" int psoap_batch_upload(psoap_chunk* h, int B, int c, const double* lwl, const double* gp, double mu_GP)
{
    if (!h || !lwl || !gp) FAIL("psoap_batch_upload: bad arguments");
    DEVICE_SCOPE(h->device);
    if (set_dev(h)) return 1;
    BatchSlot* sl = nullptr;
    if (int rc = upload_begin(h, B, c, gp, mu_GP, &sl)) return rc;
    const size_t nl = (size_t)B * c * h->N;
    memcpy(h->hLwl, lwl, sizeof(double) * nl);
    if (int rc = clear_too_fast(h, *sl)) return rc;
    HIP_TRY(hipMemcpyAsync(sl->dLwl, h->hLwl, sizeof(double) * nl, hipMemcpyHostToDevice, h->copy));
    return upload_end(h, *sl);
}

extern "C" int psoap_batch_upload_velocities(psoap_chunk* h, int B, int c, const double* vel, const double* gp,
                                             double mu_GP)
{
    if (!h || !vel || !gp) FAIL("psoap_batch_upload_velocities: bad arguments");
    if (!h->dGrid) FAIL("psoap_batch_upload_velocities: call psoap_chunk_set_grid first");
    DEVICE_SCOPE(h->device);
    if (set_dev(h)) return 1;
    BatchSlot* sl = nullptr;
    if (int rc = upload_begin(h, B, c, gp, mu_GP, &sl)) return rc;
    const size_t nv = (size_t)B * c * h->n_epochs;
    memcpy(h->hVel, vel, sizeof(double) * nv);
    hipStream_t s = h->copy;
    if (int rc = clear_too_fast(h, *sl)) return rc;
    HIP_TRY(hipMemcpyAsync(h->dVel, h->hVel, sizeof(double) * nv, hipMemcpyHostToDevice, s));
    dim3 grid((h->N + 255) / 256, B * c);
    hipLaunchKernelGGL(k_doppler_shift, grid, dim3(256), 0, s, sl->dLwl, h->dGrid, h->dEpoch, h->dVel, h->N,
                       h->n_epochs, B * c);
    HIP_TRY(hipGetLastError());
    return upload_end(h, *sl);
}

extern "C" int psoap_chunk_set_dates(psoap_chunk* h, const double* dates, int n_epochs)
{
    if (!h || !dates) FAIL("psoap_chunk_set_dates: bad arguments");
    if (!h->dGrid || n_epochs != h->n_epochs) FAIL("psoap_chunk_set_dates: call psoap_chunk_set_grid first (same n_epochs)");
    if (h->stream.open) FAIL("psoap_chunk_set_dates: the handle has an open stream (psoap_stream_close first)");
    DEVICE_SCOPE(h->device);
    if (set_dev(h)) return 1;
    HIP_TRY(hipDeviceSynchronize());
    if (h->dDates) HIP_TRY(hipFree(h->dDates));
    h->dDates = nullptr;
    HIP_TRY(hipMalloc(&h->dDates, sizeof(double) * n_epochs));
    HIP_TRY(hipMemcpy(h->dDates, dates, sizeof(double) * n_epochs, hipMemcpyHostToDevice));
    return 0;
}

static int check_orbits(int model, int B, const double* p_orb)
{
    if (model < ORB_SB1 || model > ORB_ST3) FAIL("orbit model must be 0..4 (SB1, SB2, ST1, ST2, ST3)");
    const int np = orbit_n_params(model);
    for (int b = 0; b < B; ++b) {
        const double* p = p_orb + (size_t)b * np;
        // eccentricities: orbit.py:35,191-192 assert 0 <= e < 1
        const int o = (model == ORB_SB1) ? 1 : (model == ORB_SB2) ? 2 : (model == ORB_ST1 ? 1 : 2);
        bool ok = (p[o] >= 0.0 && p[o] < 1.0);
        if (model >= ORB_ST1) {
            const int oe = (model == ORB_ST1) ? 6 : (model == ORB_ST2 ? 7 : 8);
            ok = ok && (p[oe] >= 0.0 && p[oe] < 1.0);
        }
        if (!ok) FAIL("Eccentricity must be between [0, 1)");
    }
    return 0;
}

extern "C" int psoap_batch_upload_orbits(psoap_chunk* h, int B, int model, const double* p_orb, const double* gp,
                                         double mu_GP)
{
    if (!h || !p_orb || !gp) FAIL("psoap_batch_upload_orbits: bad arguments");
    if (!h->dGrid || !h->dDates) FAIL("psoap_batch_upload_orbits: call psoap_chunk_set_grid and psoap_chunk_set_dates first");
    DEVICE_SCOPE(h->device);
    if (set_dev(h)) return 1;
    if (B < 1 || B > h->max_batch) FAIL("batch size outside [1, max_batch]");
    if (int rc = check_orbits(model, B, p_orb)) return rc;
    const int c = orbit_n_components(model), np = orbit_n_params(model);
    BatchSlot* sl = nullptr;
    if (int rc = upload_begin(h, B, c, gp, mu_GP, &sl)) return rc;
    memcpy(h->hPorb, p_orb, sizeof(double) * (size_t)B * np);
    hipStream_t s = h->copy;
    HIP_TRY(hipMemcpyAsync(h->dPorb, h->hPorb, sizeof(double) * (size_t)B * np, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemsetAsync(sl->dTooFast, 0, sizeof(int) * (size_t)h->max_batch, s));
    sl->toofast_dirty = true;
    hipLaunchKernelGGL(k_orbit_velocities, dim3((h->n_epochs + 63) / 64, B), dim3(64), 0, s, model, B, h->n_epochs,
                       h->dPorb, h->dDates, h->dVel, sl->dTooFast);
    HIP_TRY(hipGetLastError());
    dim3 grid((h->N + 255) / 256, B * c);
    hipLaunchKernelGGL(k_doppler_shift, grid, dim3(256), 0, s, sl->dLwl, h->dGrid, h->dEpoch, h->dVel, h->N,
                       h->n_epochs, B * c);
    HIP_TRY(hipGetLastError());
    return upload_end(h, *sl);
}

// stand-alone batched orbit evaluation: vel_out (B, c, n_dates)
extern "C" int psoap_orbit_velocities(int device, int model, int B, const double* p_orb, int n_dates,
                                      const double* dates, double* vel_out)
{
    if (B < 1 || n_dates < 1 || !p_orb || !dates || !vel_out) FAIL("psoap_orbit_velocities: bad arguments");
    if (int rc = check_orbits(model, B, p_orb)) return rc;
    DEVICE_SCOPE(device);
    HIP_TRY(hipSetDevice(device));
    const int c = orbit_n_components(model), np = orbit_n_params(model);
    DevBuf<double> dP, dD, dV;
    HIP_TRY(dP.alloc((size_t)B * np));
    HIP_TRY(dD.alloc(n_dates));
    HIP_TRY(dV.alloc((size_t)B * c * n_dates));
    HIP_TRY(hipMemcpy(dP, p_orb, sizeof(double) * (size_t)B * np, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dD, dates, sizeof(double) * n_dates, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_orbit_velocities, dim3((n_dates + 63) / 64, B), dim3(64), 0, 0, model, B, n_dates, dP.p, dD.p,
                       dV.p, (int*)nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(vel_out, dV, sizeof(double) * (size_t)B * c * n_dates, hipMemcpyDeviceToHost));
    return 0;
}

// ---- profiling helpers ------------------------------------------------------------------
static int prof_begin(psoap_chunk* h, hipStream_t s, int cls, double flops, double bytes)
{
    if (!h->profiling) return 0;
    size_t need = h->recs.size() * 2 + 2;
    while (h->evPool.size() < need) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        h->evPool.push_back(e);
    }
    psoap_chunk::Rec r;
    r.cls = cls;
    r.e0 = (int)h->recs.size() * 2;
    r.e1 = r.e0 + 1;
    r.flops = flops;
    r.bytes = bytes;
    h->recs.push_back(r);
    HIP_TRY(hipEventRecord(h->evPool[r.e0], s));
    return 0;
}
static int prof_end(psoap_chunk* h, hipStream_t s)
{
    if (!h->profiling) return 0;
    HIP_TRY(hipEventRecord(h->evPool[h->recs.back().e1], s));
    return 0;
}

template <int C>
static void launch_fill(psoap_chunk* h, const BatchSlot& sl, hipStream_t s, int b0, int nb, int upper_only)
{
    const int tiles = upper_only ? h->P * (h->P + 1) / 2 : h->P * h->P;
    hipLaunchKernelGGL(k_fill_sym<C>, dim3(tiles, nb), dim3(256), 0, s, h->dK + (size_t)b0 * h->mat_stride,
                       h->mat_stride, h->ld, h->N, h->P, sl.dLwl + (size_t)b0 * C * h->N,
                       sl.dGp + (size_t)b0 * 2 * C, h->dSigma, upper_only);
}

// per-matrix records of a slot's batch (uniform: every matrix shares N, fl, sigma)
static void fill_mats(const psoap_chunk* h, const BatchSlot& sl, DagMat* out)
{
    for (int b = 0; b < sl.B; ++b) {
        DagMat m{};
        m.K = h->dK + (size_t)b * h->mat_stride;
        m.R = h->dR + (size_t)b * h->Npad;
        m.Wt = h->dWt + (size_t)b * WT_STRIDE;
        m.lw = sl.dLwl + (size_t)b * sl.C * h->N;
        m.gp = sl.dGp + (size_t)b * 2 * sl.C;
        m.sigma = h->dSigma;
        m.acc = h->dAcc + (size_t)b * ACC_ROWS;
        m.N = h->N;
        m.Npad = h->Npad;
        m.P = h->P;
        m.ld = h->ld;
        out[b] = m;
    }
}

// An evaluation consumes the pending upload, if there is one; otherwise it re-evaluates the active slot.
static int promote_slot(psoap_chunk* h, const char* who)
{
    if (h->pend >= 0) {
        h->act = h->pend;
        h->pend = -1;
    }
    if (h->act < 0 || h->slot[h->act].B < 1) {
        g_err = std::string(who) + ": nothing uploaded";
        return 2;
    }
    return 0;
}

// the slot's per-matrix records on the device: written once per (slot, B, C)
static int ensure_slot_mats(psoap_chunk* h, BatchSlot& sl)
{
    if (sl.mats_B == sl.B && sl.mats_C == sl.C) return 0;
    std::vector<DagMat> mats(sl.B);
    fill_mats(h, sl, mats.data());
    HIP_TRY(hipStreamSynchronize(h->streams[0]));   // an earlier launch may still read the records
    HIP_TRY(hipMemcpy(sl.dMats, mats.data(), sizeof(DagMat) * sl.B, hipMemcpyHostToDevice));
    sl.mats_B = sl.B;
    sl.mats_C = sl.C;
    return 0;
}

// (re)build the task list of the persistent kernel when the batch size changes
static int dag_prepare(psoap_chunk* h)
{
    BatchSlot& sl = h->slot[h->act];
    if (int rc = ensure_slot_mats(h, sl)) return rc;
    if (h->plan_B == sl.B) return 0;
    if (h->P > 255) FAIL("N too large for the persistent kernel's 8-bit block-row indices (N <= 32640)");
    HIP_TRY(hipStreamSynchronize(h->streams[0]));
    // PSOAP_DAG_SCHEME=0|1 pins the split scheme (experiments); default: automatic
    const char* env_scheme = getenv("PSOAP_DAG_SCHEME");
    const std::vector<int> Ps((size_t)sl.B, h->P);
    const int workers = dag_pick_workers(dag_batch_flops(Ps), h->P, h->n_cus, h->dag_grid, (int)Ps.size());
    // (PSOAP_FIXED_PLAN=1: the task structure of a stream lane for every matrix, whatever the batch -- dag_fixed_plan)
    DagPlan plan = dag_build_tasks(Ps, workers, env_scheme ? atoi(env_scheme) : -1, 0, 0,
                                   dag_fixed_plan() ? dag_nominal_share(h->dag_grid - 1) : 0);
    if ((size_t)plan.n_ctrs + 4 + (plan.tasks.size() + 31) / 32 + 1 > h->arrive_cap)
        FAIL("internal: arrival counter capacity exceeded");
    if (plan.tasks.size() > h->tasks_cap) {
        if (h->dTasks) HIP_TRY(hipFree(h->dTasks));
        if (h->dOrder) HIP_TRY(hipFree(h->dOrder));
        if (h->dDep) HIP_TRY(hipFree(h->dDep));
        h->dTasks = nullptr;
        h->dOrder = h->dDep = nullptr;
        h->tasks_cap = 0;
        HIP_TRY(hipMalloc(&h->dTasks, sizeof(DagTask) * plan.tasks.size()));
        HIP_TRY(hipMalloc(&h->dOrder, sizeof(unsigned int) * plan.tasks.size()));
        HIP_TRY(hipMalloc(&h->dDep, sizeof(unsigned int) * plan.tasks.size()));
        h->tasks_cap = plan.tasks.size();
    }
    h->plan_pool = !plan.order.empty();
    if (h->plan_pool) {
        HIP_TRY(hipMemcpy(h->dOrder, plan.order.data(), sizeof(unsigned int) * plan.order.size(), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(h->dDep, plan.dep.data(), sizeof(unsigned int) * plan.dep.size(), hipMemcpyHostToDevice));
        memcpy(h->plan_n_main, plan.n_main, sizeof h->plan_n_main);
    }
    if (plan.n_slots > h->ws_cap) {
        if (h->dWs) HIP_TRY(hipFree(h->dWs));
        h->dWs = nullptr;
        h->ws_cap = 0;
        HIP_TRY(hipMalloc(&h->dWs, sizeof(double) * NB * NB * (size_t)plan.n_slots));
        h->ws_cap = plan.n_slots;
    }
    HIP_TRY(hipMemcpy(h->dTasks, plan.tasks.data(), sizeof(DagTask) * plan.tasks.size(), hipMemcpyHostToDevice));
    h->plan_B = sl.B;
    h->plan_workers = workers;
    h->plan_scheme = plan.scheme;
    h->plan_tasks = (unsigned int)plan.tasks.size();
    h->plan_ctrs = plan.n_ctrs;
    h->plan_slots = plan.n_slots;
    h->plan_queues = plan.queues;
    return 0;
}

// Many small matrices: one workgroup per matrix (solo_kernel.hpp) instead of the dependency graph.  PSOAP_SOLO=1 / 0 forces /
// forbids it; otherwise from PSOAP_SOLO_MIN matrices on (default: solo_min_default, from the measured table of round 6)
// while the largest matrix has at most PSOAP_SOLO_MAX_P block rows.
static bool solo_wanted(int n_mats, int Pmax)
{
    // (read at every launch: the tests and tools switch it inside one process)
    const char* e = getenv("PSOAP_SOLO");
    if (e && e[0] == '1') return true;
    if (e && e[0] == '0') return false;
    const int min_mats = getenv("PSOAP_SOLO_MIN") ? atoi(getenv("PSOAP_SOLO_MIN")) : 0x7fffffff;
    const int max_p = getenv("PSOAP_SOLO_MAX_P") ? atoi(getenv("PSOAP_SOLO_MAX_P")) : 24;
    return n_mats >= min_mats && Pmax <= max_p;
}
template <class... Args>
static void launch_solo(int C, int grid, hipStream_t s, Args... args)
{
    if (C == 1) hipLaunchKernelGGL(k_chol_solo<1>, dim3(grid), dim3(GEMM_THREADS), GEMM_LDS_BYTES, s, args...);
    else if (C == 2) hipLaunchKernelGGL(k_chol_solo<2>, dim3(grid), dim3(GEMM_THREADS), GEMM_LDS_BYTES, s, args...);
    else hipLaunchKernelGGL(k_chol_solo<3>, dim3(grid), dim3(GEMM_THREADS), GEMM_LDS_BYTES, s, args...);
}

// One persistent launch for the whole batched factorisation (dag_kernel.hpp).
static int eval_dag(psoap_chunk* h)
{
    BatchSlot& sl = h->slot[h->act];
    const int B = sl.B, C = sl.C, N = h->N, P = h->P;
    hipStream_t s = h->streams[0];
    h->recs.clear();
    if (int rc = dag_prepare(h)) return rc;
    HIP_TRY(hipStreamWaitEvent(s, sl.evUpload, 0));
    // no fill kernel: the DAG kernel evaluates the covariance tiles on the fly (dag_store_updated)
    if (prof_begin(h, s, PSOAP_K_MISC, 0.0, 0.0)) return 1;
    hipLaunchKernelGGL(k_init_rhs, dim3((h->Npad + 255) / 256, B), dim3(256), 0, s, h->dR, h->Npad, N, h->dFl,
                       sl.mu, h->dAcc);
    HIP_TRY(hipGetLastError());
    // (flags, arrival counters and -- behind them -- the taken bitmap of the ready-only hand-out: one memset)
    const size_t taken_off = h->arrive_off + sizeof(int) * ((size_t)h->plan_ctrs + 4);
    HIP_TRY(hipMemsetAsync(h->dDag, 0, taken_off + sizeof(unsigned int) * (((size_t)h->plan_tasks + 31) / 32 + 1), s));
    // PSOAP_DEBUG_POISON (tools/soak_batch_perm.py; bit 0: the matrices, bit 1: the mailboxes, bit 2: the partial-tile
    // workspace, bit 3: the block records of the accumulators): NaN patterns in whatever the launch must write before it
    // reads -- a task that reads ahead of its producer then returns NaN instead of the previous launch's (possibly
    // identical) bits.  Everything poisoned here is written in full by the launch: upper-triangle tiles incl. their
    // identity padding, the mailbox slots a follower reads, every slot a PART chain uses, one record per block row.
    static const int poison = getenv("PSOAP_DEBUG_POISON") ? atoi(getenv("PSOAP_DEBUG_POISON")) : 0;
    if (poison & 1) HIP_TRY(hipMemsetAsync(h->dK, 0xFF, sizeof(double) * h->mat_stride * (size_t)B, s));
    if (poison & 2)
        for (int b = 0; b < B; ++b)
            HIP_TRY(hipMemsetAsync(h->dWt + (size_t)b * WT_STRIDE + (size_t)2 * NB * NB, 0xFF, sizeof(double) * MB_DOUBLES, s));
    if ((poison & 4) && h->dWs && h->plan_slots) HIP_TRY(hipMemsetAsync(h->dWs, 0xFF, sizeof(double) * NB * NB * (size_t)h->plan_slots, s));
    if (poison & 8) HIP_TRY(hipMemsetAsync(h->dAcc, 0xFF, sizeof(MatAcc) * ACC_ROWS * (size_t)B, s));
    if (prof_end(h, s)) return 1;
    const long long tasks = h->plan_tasks;
    const int grid_all = (int)(tasks < h->plan_workers ? tasks : h->plan_workers);
    // executed MFMA flops: left-looking updates + strip solves, full 128^3 tiles
    double fl = 0.0;
    for (int q = 0; q < P; ++q) fl += 2.0 * NB * NB * ((double)q * NB * (P - q) + (double)NB * (P - q - 1));
    if (prof_begin(h, s, PSOAP_K_DAG, fl * B, 0.0)) return 1;
    if (solo_wanted(B, P)) {
        const int grid = B < h->dag_grid ? B : h->dag_grid;
        launch_solo(C, grid, s, (const DagMat*)sl.dMats, (const unsigned int*)nullptr, B, reinterpret_cast<SoloCtl*>(h->dDag));
    } else {
        MatFlags* fl_ = reinterpret_cast<MatFlags*>(h->dDag + sizeof(DagCtl));
        DagCtl* ctl_ = reinterpret_cast<DagCtl*>(h->dDag);
#define PSOAP_LAUNCH_DAG(CC, LAT, WPE)                                                                           \
    hipLaunchKernelGGL((k_chol_dag<CC, false, LAT, false, WPE>), dim3(grid), dim3(GEMM_THREADS), GEMM_LDS_BYTES, s, \
                       sl.dMats, h->dTasks, h->plan_queues, fl_, reinterpret_cast<int*>(h->dDag + h->arrive_off),   \
                       h->dWs, ctl_, h->dTlog, DagAug{P, 0, 0, nullptr}, StreamArgs{}, pool_)
        DagPool pool_{};
        if (h->plan_pool) {
            pool_.order = h->dOrder;
            pool_.dep = h->dDep;
            pool_.taken = reinterpret_cast<unsigned int*>(h->dDag + taken_off);
            memcpy(pool_.n_main, h->plan_n_main, sizeof pool_.n_main);
        }
        const bool lat = h->plan_scheme >= 1;
        const int grid = (lat && grid_all > 2 * h->n_cus) ? 2 * h->n_cus : grid_all;     // (the LAT kernels: two per compute unit)
        // at most one workgroup per compute unit (single evaluations: dag_pick_workers): the kernels compiled for one wave
        // per SIMD -- 512 registers per lane, nothing of the chain phases in scratch memory
        const bool wide = lat && grid <= h->n_cus && !(getenv("PSOAP_DAG_WIDE") && getenv("PSOAP_DAG_WIDE")[0] == '0');
        if (C == 1) { if (wide) PSOAP_LAUNCH_DAG(1, true, 1); else if (lat) PSOAP_LAUNCH_DAG(1, true, 2); else PSOAP_LAUNCH_DAG(1, false, DAG_WPE_TP); }
        else if (C == 2) { if (wide) PSOAP_LAUNCH_DAG(2, true, 1); else if (lat) PSOAP_LAUNCH_DAG(2, true, 2); else PSOAP_LAUNCH_DAG(2, false, DAG_WPE_TP); }
        else { if (wide) PSOAP_LAUNCH_DAG(3, true, 1); else if (lat) PSOAP_LAUNCH_DAG(3, true, 2); else PSOAP_LAUNCH_DAG(3, false, DAG_WPE_TP); }
#undef PSOAP_LAUNCH_DAG
    }
    HIP_TRY(hipGetLastError());
    if (prof_end(h, s)) return 1;
    if (prof_begin(h, s, PSOAP_K_MISC, 0.0, 0.0)) return 1;
    hipLaunchKernelGGL(k_finalize, dim3((B + 63) / 64), dim3(64), 0, s, h->dAcc, h->dOut, B, sl.dTooFast, P);
    HIP_TRY(hipGetLastError());
    if (prof_end(h, s)) return 1;
    HIP_TRY(hipEventRecord(sl.evEval, s));
    HIP_TRY(hipEventRecord(h->evLast, s));
    h->last_recorded = true;
    HIP_TRY(hipMemcpyAsync(h->hOut, h->dOut, sizeof(double) * B, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(h->hDagErr, h->dDag + offsetof(DagCtl, error), 6 * sizeof(unsigned int),
                           hipMemcpyDeviceToHost, s));
    h->last_path = 1;
    g_share.dag_launches += 1;
    return 0;
}

static int eval_staged(psoap_chunk* h);

extern "C" int psoap_batch_eval(psoap_chunk* h)
{
    if (!h) FAIL("psoap_batch_eval: null handle");
    DEVICE_SCOPE(h->device);
    if (set_dev(h)) return 1;
    if (int rc = promote_slot(h, "psoap_batch_eval")) return rc;
    h->last_group = nullptr;
    if (int rc = handle_lock(h)) return rc;      // released by the fetch / sync that sees the evaluation complete
    // the persistent kernel indexes block rows with 8 bits (beyond N = 32640: the staged path), and it is not used where
    // it cannot be made safe among several processes (share_wants_staged)
    int rc;
    if (h->mode == 1 && h->P <= 255 && !share_wants_staged(h->device)) {
        rc = eval_dag(h);
    } else {
        if (h->mode == 1 && h->P <= 255) g_share.staged_policy += 1;
        rc = eval_staged(h);
    }
    if (rc) handle_unlock(h);
    return rc;
}

// The staged path: three kernels per block row, synchronised by kernel boundaries.
static int eval_staged(psoap_chunk* h)
{
    h->last_path = 0;
    BatchSlot& sl = h->slot[h->act];
    const int B = sl.B, C = sl.C, N = h->N, P = h->P;
    const int G = h->profiling ? 1 : (h->groups < B ? h->groups : B);
    for (int g = 1; g < G; ++g)
        if (!h->streams[g]) HIP_TRY(hipStreamCreateWithFlags(&h->streams[g], hipStreamNonBlocking));
    h->recs.clear();
    int gb0[MAX_GROUPS + 1];
    for (int g = 0; g <= G; ++g) gb0[g] = (int)((long long)B * g / G);

    // stage 0 per group: wait for the upload, fill K (upper tiles), r = fl - mu, clear accumulators
    for (int g = 0; g < G; ++g) {
        hipStream_t s = h->streams[g];
        const int b0 = gb0[g], nb = gb0[g + 1] - gb0[g];
        HIP_TRY(hipStreamWaitEvent(s, sl.evUpload, 0));
        // the shared workspaces: after the previous evaluation of this HANDLE (it normally ran on the other slot, and
        // the stream-group boundaries move with B, so the slot's own evEval says nothing about them)
        if (g != 0 && h->last_recorded) HIP_TRY(hipStreamWaitEvent(s, h->evLast, 0));
        const double fbytes = (double)nb * (4.0 * N * (N + 1.0) + 8.0 * (C + 1.0) * N);
        if (prof_begin(h, s, PSOAP_K_FILL, 0.0, fbytes)) return 1;
        if (C == 1) launch_fill<1>(h, sl, s, b0, nb, 1);
        else if (C == 2) launch_fill<2>(h, sl, s, b0, nb, 1);
        else launch_fill<3>(h, sl, s, b0, nb, 1);
        HIP_TRY(hipGetLastError());
        if (prof_end(h, s)) return 1;
        if (prof_begin(h, s, PSOAP_K_MISC, 0.0, 0.0)) return 1;
        hipLaunchKernelGGL(k_init_rhs, dim3((h->Npad + 255) / 256, nb), dim3(256), 0, s,
                           h->dR + (size_t)b0 * h->Npad, h->Npad, N, h->dFl, sl.mu, h->dAcc + (size_t)b0 * ACC_ROWS);
        HIP_TRY(hipGetLastError());
        if (prof_end(h, s)) return 1;
    }
    // panels, round-robin over the stream groups so their phases interleave on the device
    for (int p = 0; p < P; ++p) {
        const int k0 = p * NB;
        const int ntile = P - p;
        for (int g = 0; g < G; ++g) {
            hipStream_t s = h->streams[g];
            const int b0 = gb0[g], nb = gb0[g + 1] - gb0[g];
            double* Kg = h->dK + (size_t)b0 * h->mat_stride;
            if (p > 0) {
                if (prof_begin(h, s, PSOAP_K_PANEL_UPDATE, 2.0 * NB * NB * (double)k0 * ntile * nb, 0.0)) return 1;
                hipLaunchKernelGGL(k_panel_update, dim3(ntile, nb), dim3(GEMM_THREADS), GEMM_LDS_BYTES, s, Kg,
                                   h->mat_stride, h->ld, k0);
                HIP_TRY(hipGetLastError());
                if (prof_end(h, s)) return 1;
            }
            if (prof_begin(h, s, PSOAP_K_POTRF, 0.0, 0.0)) return 1;
            hipLaunchKernelGGL(k_potrf_diag, dim3(nb), dim3(512), 0, s, Kg, h->mat_stride, h->ld, k0,
                               h->dWt + (size_t)b0 * WT_STRIDE, h->dR + (size_t)b0 * h->Npad, h->Npad, h->dAcc + (size_t)b0 * ACC_ROWS,
                               (size_t)WT_STRIDE);
            HIP_TRY(hipGetLastError());
            if (prof_end(h, s)) return 1;
            if (ntile > 1) {
                if (prof_begin(h, s, PSOAP_K_TRSM, 2.0 * NB * NB * (double)NB * (ntile - 1) * nb, 0.0)) return 1;
                hipLaunchKernelGGL(k_trsm_strip, dim3(ntile - 1, nb), dim3(GEMM_THREADS), GEMM_LDS_BYTES, s, Kg,
                                   h->mat_stride, h->ld, k0, h->dWt + (size_t)b0 * WT_STRIDE,
                                   h->dR + (size_t)b0 * h->Npad, h->Npad, (size_t)WT_STRIDE);
                HIP_TRY(hipGetLastError());
                if (prof_end(h, s)) return 1;
            }
        }
    }
    // finalize per group; group 0's stream gathers the others and does the D2H
    for (int g = 0; g < G; ++g) {
        hipStream_t s = h->streams[g];
        const int b0 = gb0[g], nb = gb0[g + 1] - gb0[g];
        if (prof_begin(h, s, PSOAP_K_MISC, 0.0, 0.0)) return 1;
        hipLaunchKernelGGL(k_finalize, dim3((nb + 63) / 64), dim3(64), 0, s, h->dAcc + (size_t)b0 * ACC_ROWS, h->dOut + b0, nb,
                           sl.dTooFast + b0, P);
        HIP_TRY(hipGetLastError());
        if (prof_end(h, s)) return 1;
        if (g != 0) {
            HIP_TRY(hipEventRecord(h->evDone[g], s));
            HIP_TRY(hipStreamWaitEvent(h->streams[0], h->evDone[g], 0));
        }
    }
    HIP_TRY(hipEventRecord(sl.evEval, h->streams[0]));
    HIP_TRY(hipEventRecord(h->evLast, h->streams[0]));
    h->last_recorded = true;
    HIP_TRY(hipMemcpyAsync(h->hOut, h->dOut, sizeof(double) * B, hipMemcpyDeviceToHost, h->streams[0]));
    return 0;
}

static int collect_timings(psoap_chunk* h)
{
    psoap_timings t;
    memset(&t, 0, sizeof t);
    if (h->profiling && !h->recs.empty()) {
        for (auto& r : h->recs) {
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, h->evPool[r.e0], h->evPool[r.e1]));
            t.ms[r.cls] += ms;
            t.launches[r.cls] += 1;
            t.flops[r.cls] += r.flops;
            t.bytes[r.cls] += r.bytes;
        }
        float tot = 0.f;
        HIP_TRY(hipEventElapsedTime(&tot, h->evPool[h->recs.front().e0], h->evPool[h->recs.back().e1]));
        t.total_ms = tot;
    }
    h->last = t;
    return 0;
}

// ---- several chunks, one launch ------------------------------------------------------------------
// The uploaded batches of several chunk handles (one device, one component count) are factored by ONE
// launch of the persistent kernel over the heterogeneous batch: every matrix brings its own size, data
// and storage (DagMat), the task list walks all of them together, and the matrices of all chunks hide
// each other's dependency chains -- what a workload of many small chunks needs to fill the device.
struct psoap_group {
    int device = 0;
    std::vector<psoap_chunk*> hs;
    hipStream_t stream = nullptr;
    hipEvent_t evDone = nullptr;
    unsigned char* dDag = nullptr;
    size_t dag_cap = 0, dag_bytes = 0, arrive_off = 0;
    DagMat* dMats = nullptr;
    size_t mats_cap = 0;
    DagTask* dTasks = nullptr;
    unsigned int* dOrder = nullptr;   // DagPool: order[], dep[] (tasks_cap entries each)
    unsigned int* dDep = nullptr;
    unsigned int n_main[DAG_QUEUES] = {};
    bool pool = false;
    size_t taken_off = 0;
    size_t tasks_cap = 0;
    double* dWs = nullptr;
    size_t ws_cap = 0;
    std::vector<int> key;      // B of every handle, then C: the plan is rebuilt when it changes
    std::vector<int> acts;     // proposal slot of every handle the records in dMats point into
    long long plan_builds = 0, record_refreshes = 0;   // psoap_group_stats
    DagQueues queues{};
    long long n_tasks = 0;
    int total_B = 0;
    int scheme = 0;
    int workers = 0;
};

extern "C" int psoap_group_destroy(psoap_group* g);
static std::mutex g_groups_mu;
static std::set<psoap_group*> g_live_groups;      // groups that exist: a handle's last_group is only followed while it is in here

extern "C" int psoap_group_create(psoap_group** out, psoap_chunk* const* handles, int n)
{
    if (!out || !handles || n < 1) FAIL("psoap_group_create: bad arguments");
    for (int k = 0; k < n; ++k) {
        if (!handles[k]) FAIL("psoap_group_create: null handle");
        if (handles[k]->device != handles[0]->device) FAIL("psoap_group_create: all chunks must live on one device");
        if (handles[k]->P > 255) FAIL("psoap_group_create: N too large for the persistent kernel (N <= 32640)");
    }
    *out = nullptr;
    DEVICE_SCOPE(handles[0]->device);
    HIP_TRY(hipSetDevice(handles[0]->device));
    psoap_group* g = new psoap_group();
    g->device = handles[0]->device;
    g->hs.assign(handles, handles + n);
    hipError_t e = hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&g->evDone, hipEventDisableTiming);
    if (e != hipSuccess) {
        (void)psoap_group_destroy(g);
        g_err = std::string("psoap_group_create: ") + hipGetErrorString(e);
        return 1;
    }
    {
        std::lock_guard<std::mutex> lk(g_groups_mu);
        g_live_groups.insert(g);
    }
    *out = g;
    return 0;
}

extern "C" int psoap_group_destroy(psoap_group* g)
{
    if (!g) return 0;
    DEVICE_SCOPE_DESTROY(g->device);
    (void)hipSetDevice(g->device);
    (void)hipDeviceSynchronize();
    {   // (a member's fetch must not come back to a group that is gone: settle_evaluation asks the registry)
        std::lock_guard<std::mutex> lk(g_groups_mu);
        g_live_groups.erase(g);
    }
    (void)hipFree(g->dDag); (void)hipFree(g->dMats); (void)hipFree(g->dTasks); (void)hipFree(g->dWs);
    (void)hipFree(g->dOrder); (void)hipFree(g->dDep);
    if (g->stream) (void)hipStreamDestroy(g->stream);
    if (g->evDone) (void)hipEventDestroy(g->evDone);
    delete g;
    return 0;
}

// Launch the uploaded batches of all member handles; afterwards psoap_batch_fetch on each handle returns
// its results (every handle's stream waits for the group launch).
static int group_eval_locked(psoap_group* g, bool promote);

extern "C" int psoap_group_eval(psoap_group* g)
{
    if (!g) FAIL("psoap_group_eval: null group");
    HIP_TRY(hipSetDevice(g->device));
    for (psoap_chunk* h : g->hs)                      // each member's fetch releases its own reference
        if (int rc = handle_lock(h)) {
            for (psoap_chunk* k : g->hs) handle_unlock(k);
            return rc;
        }
    int rc = 0;
    if (share_wants_staged(g->device)) {
        // several processes on the device and no safe persistent launch: every member by the staged path, one after the other
        for (psoap_chunk* h : g->hs) {
            if (!rc) rc = promote_slot(h, "psoap_group_eval (every member needs an uploaded batch)");
            h->last_group = nullptr;
            if (!rc) g_share.staged_policy += 1;
            if (!rc) rc = eval_staged(h);
        }
    } else {
        rc = group_eval_locked(g, true);
    }
    if (rc)
        for (psoap_chunk* h : g->hs) handle_unlock(h);
    return rc;
}

// promote == false: the launch of the slots that are active now, again (settle_evaluation) -- an upload queued meanwhile
// for the NEXT step stays pending
static int group_eval_locked(psoap_group* g, bool promote)
{
    std::vector<int> key, acts;
    int total = 0;
    if (promote)
        for (psoap_chunk* h : g->hs)
            if (int rc = promote_slot(h, "psoap_group_eval (every member needs an uploaded batch)")) return rc;
    const int C = g->hs[0]->slot[g->hs[0]->act].C;
    for (psoap_chunk* h : g->hs) {
        const BatchSlot& sl = h->slot[h->act];
        if (sl.C != C) FAIL("psoap_group_eval: all members must use the same number of components");
        key.push_back(sl.B);
        acts.push_back(h->act);
        total += sl.B;
    }
    key.push_back(C);
    if (total > 65535) FAIL("psoap_group_eval: more than 65535 matrices in one launch");
    // The task list depends on the batch sizes only.  (Round 2 had the proposal slots in this key as well: in an
    // upload / eval loop they flip every step, so every step rebuilt the plan behind a hipDeviceSynchronize -- which
    // also waited for the copy stream and undid the two-slot upload pipeline.)
    if (key != g->key) {
        HIP_TRY(hipDeviceSynchronize());
        std::vector<int> Ps;
        for (psoap_chunk* h : g->hs)
            for (int b = 0; b < h->slot[h->act].B; ++b) Ps.push_back(h->P);
        const char* env_scheme = getenv("PSOAP_DAG_SCHEME");
        int Pmax = 0;
        for (int P : Ps) Pmax = P > Pmax ? P : Pmax;
        g->workers = dag_pick_workers(dag_batch_flops(Ps), Pmax, g->hs[0]->n_cus, g->hs[0]->dag_grid, (int)Ps.size());
        DagPlan plan = dag_build_tasks(Ps, g->workers, env_scheme ? atoi(env_scheme) : -1, 0, 0,
                                       dag_fixed_plan() ? dag_nominal_share(g->hs[0]->dag_grid - 1) : 0);
        if ((size_t)total > g->mats_cap) {
            if (g->dMats) HIP_TRY(hipFree(g->dMats));
            g->dMats = nullptr;
            HIP_TRY(hipMalloc(&g->dMats, sizeof(DagMat) * (size_t)total));
            g->mats_cap = (size_t)total;
        }
        if (plan.tasks.size() > g->tasks_cap) {
            if (g->dTasks) HIP_TRY(hipFree(g->dTasks));
            if (g->dOrder) HIP_TRY(hipFree(g->dOrder));
            if (g->dDep) HIP_TRY(hipFree(g->dDep));
            g->dTasks = nullptr;
            g->dOrder = g->dDep = nullptr;
            g->tasks_cap = 0;
            HIP_TRY(hipMalloc(&g->dTasks, sizeof(DagTask) * plan.tasks.size()));
            HIP_TRY(hipMalloc(&g->dOrder, sizeof(unsigned int) * plan.tasks.size()));
            HIP_TRY(hipMalloc(&g->dDep, sizeof(unsigned int) * plan.tasks.size()));
            g->tasks_cap = plan.tasks.size();
        }
        g->pool = !plan.order.empty();
        if (g->pool) {
            HIP_TRY(hipMemcpy(g->dOrder, plan.order.data(), sizeof(unsigned int) * plan.order.size(), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(g->dDep, plan.dep.data(), sizeof(unsigned int) * plan.dep.size(), hipMemcpyHostToDevice));
            memcpy(g->n_main, plan.n_main, sizeof g->n_main);
        }
        if ((size_t)plan.n_slots + 1 > g->ws_cap) {
            if (g->dWs) HIP_TRY(hipFree(g->dWs));
            g->dWs = nullptr;
            HIP_TRY(hipMalloc(&g->dWs, sizeof(double) * NB * NB * ((size_t)plan.n_slots + 1)));
            g->ws_cap = (size_t)plan.n_slots + 1;
        }
        g->arrive_off = sizeof(DagCtl) + sizeof(MatFlags) * (size_t)total;
        g->taken_off = g->arrive_off + sizeof(int) * ((size_t)plan.n_ctrs + 4);
        g->dag_bytes = g->taken_off + sizeof(unsigned int) * ((plan.tasks.size() + 31) / 32 + 1);
        if (g->dag_bytes > g->dag_cap) {
            if (g->dDag) HIP_TRY(hipFree(g->dDag));
            g->dDag = nullptr;
            HIP_TRY(hipMalloc(&g->dDag, g->dag_bytes));
            g->dag_cap = g->dag_bytes;
        }
        HIP_TRY(hipMemcpy(g->dTasks, plan.tasks.data(), sizeof(DagTask) * plan.tasks.size(), hipMemcpyHostToDevice));
        g->queues = plan.queues;
        g->scheme = plan.scheme;
        g->n_tasks = (long long)plan.tasks.size();
        g->total_B = total;
        g->key = key;
        g->acts.clear();
        ++g->plan_builds;
    }
    hipStream_t s = g->stream;
    // Only the matrix records (storage, size, proposal arrays) depend on the slots: every slot keeps its own on the
    // device (ensure_slot_mats: written once per slot and batch size), and the group's array is refreshed from them by
    // device-to-device copies queued on the group's stream -- behind the previous launch, no host synchronisation.
    if (acts != g->acts) {
        size_t b0 = 0;
        for (psoap_chunk* h : g->hs) {
            BatchSlot& sl = h->slot[h->act];
            if (int rc = ensure_slot_mats(h, sl)) return rc;
            HIP_TRY(hipMemcpyAsync(g->dMats + b0, sl.dMats, sizeof(DagMat) * (size_t)sl.B, hipMemcpyDeviceToDevice, s));
            b0 += (size_t)sl.B;
        }
        g->acts = acts;
        ++g->record_refreshes;
    }
    for (psoap_chunk* h : g->hs) {
        const BatchSlot& sl = h->slot[h->act];
        HIP_TRY(hipStreamWaitEvent(s, sl.evUpload, 0));
        hipLaunchKernelGGL(k_init_rhs, dim3((h->Npad + 255) / 256, sl.B), dim3(256), 0, s, h->dR, h->Npad, h->N, h->dFl,
                           sl.mu, h->dAcc);
        h->recs.clear();
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemsetAsync(g->dDag, 0, g->dag_bytes, s));
    int Pmax_g = 0;
    for (psoap_chunk* h : g->hs) Pmax_g = h->P > Pmax_g ? h->P : Pmax_g;
    if (solo_wanted(total, Pmax_g)) {
        const int grid = total < g->hs[0]->dag_grid ? total : g->hs[0]->dag_grid;
        launch_solo(C, grid, s, (const DagMat*)g->dMats, (const unsigned int*)nullptr, total, reinterpret_cast<SoloCtl*>(g->dDag));
    } else {
        const int workers = (g->scheme >= 1 && g->workers > 2 * g->hs[0]->n_cus) ? 2 * g->hs[0]->n_cus : g->workers;
        const int grid = (int)(g->n_tasks < workers ? g->n_tasks : workers);
        MatFlags* fl_ = reinterpret_cast<MatFlags*>(g->dDag + sizeof(DagCtl));
        DagCtl* ctl_ = reinterpret_cast<DagCtl*>(g->dDag);
#define PSOAP_LAUNCH_GROUP(CC, LAT, WPE)                                                                        \
    hipLaunchKernelGGL((k_chol_dag<CC, false, LAT, false, WPE>), dim3(grid), dim3(GEMM_THREADS), GEMM_LDS_BYTES, s, \
                       g->dMats, g->dTasks, g->queues, fl_, reinterpret_cast<int*>(g->dDag + g->arrive_off),        \
                       g->dWs, ctl_, (unsigned long long*)nullptr, DagAug{0, 0, 0, nullptr}, StreamArgs{}, pool_)
        DagPool pool_{};
        if (g->pool) {
            pool_.order = g->dOrder;
            pool_.dep = g->dDep;
            pool_.taken = reinterpret_cast<unsigned int*>(g->dDag + g->taken_off);
            memcpy(pool_.n_main, g->n_main, sizeof pool_.n_main);
        }
        const bool lat = g->scheme >= 1;
        const bool wide = lat && grid <= g->hs[0]->n_cus && !(getenv("PSOAP_DAG_WIDE") && getenv("PSOAP_DAG_WIDE")[0] == '0');
        if (C == 1) { if (wide) PSOAP_LAUNCH_GROUP(1, true, 1); else if (lat) PSOAP_LAUNCH_GROUP(1, true, 2); else PSOAP_LAUNCH_GROUP(1, false, DAG_WPE_TP); }
        else if (C == 2) { if (wide) PSOAP_LAUNCH_GROUP(2, true, 1); else if (lat) PSOAP_LAUNCH_GROUP(2, true, 2); else PSOAP_LAUNCH_GROUP(2, false, DAG_WPE_TP); }
        else { if (wide) PSOAP_LAUNCH_GROUP(3, true, 1); else if (lat) PSOAP_LAUNCH_GROUP(3, true, 2); else PSOAP_LAUNCH_GROUP(3, false, DAG_WPE_TP); }
#undef PSOAP_LAUNCH_GROUP
    }
    HIP_TRY(hipGetLastError());
    for (psoap_chunk* h : g->hs) {
        const BatchSlot& sl = h->slot[h->act];
        hipLaunchKernelGGL(k_finalize, dim3((sl.B + 63) / 64), dim3(64), 0, s, h->dAcc, h->dOut, sl.B, sl.dTooFast, h->P);
        HIP_TRY(hipEventRecord(sl.evEval, s));
        HIP_TRY(hipEventRecord(h->evLast, s));
        h->last_recorded = true;
        HIP_TRY(hipMemcpyAsync(h->hOut, h->dOut, sizeof(double) * sl.B, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(h->hDagErr, g->dDag + offsetof(DagCtl, error), 6 * sizeof(unsigned int),
                               hipMemcpyDeviceToHost, s));
        h->last_path = 1;
        h->last_group = g;
    }
    g_share.dag_launches += 1;
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(g->evDone, s));
    for (psoap_chunk* h : g->hs) HIP_TRY(hipStreamWaitEvent(h->streams[0], g->evDone, 0));
    return 0;
}

extern "C" int psoap_group_stats(psoap_group* g, long long* plan_builds, long long* record_refreshes)
{
    if (!g) FAIL("psoap_group_stats: null group");
    if (plan_builds) *plan_builds = g->plan_builds;
    if (record_refreshes) *record_refreshes = g->record_refreshes;
    return 0;
}


// ---- streamed evaluation ---------------------------------------------------------------------------
// The reference issues one iteration after another (psoap/sample_parallel.py:434-438: the sampler's loop; :193 the
// likelihood call inside it).  psoap_stream_* keeps ONE launch of the persistent kernel resident across those
// iterations: submit() hands proposals to free lanes, fetch() returns their lnprob; see dag_kernel.hpp.
static int stream_measure_launch(psoap_chunk* h);

static int stream_launch(psoap_chunk* h)
{
    StreamState& st = h->stream;
    hipStream_t s = h->streams[0];
    HIP_TRY(hipMemsetAsync(st.dDev, 0, 64, s));      // stop, opens (no workgroup of an earlier launch is left: same stream)
    HIP_TRY(hipEventRecord(st.evStart, s));
    StreamArgs a{};
    a.lanes = st.dLanes;
    a.dev = st.dDev;
    a.host = st.hHost;
    a.h_lw = st.hLw;
    a.h_gp = st.hGp;
    a.fl = h->dFl;
    a.grid = h->dGrid;                 // (nullptr without psoap_chunk_set_grid / _set_dates: ln-wavelength submissions only)
    a.epoch = h->dEpoch;
    a.dates = h->dDates;
    a.n_epochs = h->n_epochs;
    a.h_stride = (int)st.h_stride;
    a.n_lanes = (unsigned int)st.lanes;
    a.n_tasks = st.n_tasks;
    a.ctrs_per_lane = st.ctrs_per_lane;
    a.slots_per_lane = st.slots_per_lane;
    a.C = st.C;
    a.N = h->N;
    a.idle_ticks = (unsigned long long)(st.idle_ms * 1e5);      // s_memrealtime: 100 MHz
    {
        // ticks (10 ns) between two block rows of one lane; PSOAP_STREAM_GATE_US overrides (0: lanes strictly in turn)
        const char* e = getenv("PSOAP_STREAM_GATE_US");
        const double us = e ? atof(e) : 250.0;
        a.gate = (st.scheme == 0 && DAG_TILE_DEPS && us > 0.0) ? (unsigned int)(us * 100.0) : 0u;
    }
    a.tlog_cap = st.tlog_cap;
    MatFlags* fl_ = reinterpret_cast<MatFlags*>(st.dDag + sizeof(DagCtl));
    DagCtl* ctl_ = reinterpret_cast<DagCtl*>(st.dDag);
    // (the stream kernels are compiled for two workgroups per compute unit, also in a -DPSOAP_WPE3 build: with three, hipcc's
    // code for them shows the exec-restore defect of psoap_amd/asmcheck.py and the build refuses it)
    const int grid = h->dag_grid > 2 * h->n_cus ? 2 * h->n_cus : h->dag_grid;
#define PSOAP_LAUNCH_STREAM(CC, LAT, WPE)                                                                         \
    hipLaunchKernelGGL((k_chol_dag<CC, false, LAT, true, WPE>), dim3(grid), dim3(GEMM_THREADS), GEMM_LDS_BYTES, s, \
                       st.dMats, st.dTasks, st.queues, fl_, reinterpret_cast<int*>(st.dDag + st.arrive_off), st.dWs, \
                       ctl_, st.dTlog, DagAug{h->P, 0, 0, nullptr}, a, DagPool{})
    const bool lat = st.scheme >= 1;
    if (st.C == 1) { if (lat) PSOAP_LAUNCH_STREAM(1, true, 2); else PSOAP_LAUNCH_STREAM(1, false, 2); }
    else if (st.C == 2) { if (lat) PSOAP_LAUNCH_STREAM(2, true, 2); else PSOAP_LAUNCH_STREAM(2, false, 2); }
    else { if (lat) PSOAP_LAUNCH_STREAM(3, true, 2); else PSOAP_LAUNCH_STREAM(3, false, 2); }
#undef PSOAP_LAUNCH_STREAM
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(st.evExit, s));
    st.launched = true;
    ++st.launches;
    return 0;
}

// the resident launch is there, or comes back (it leaves by itself when nothing was in flight for idle_ms)
static int stream_ensure_running(psoap_chunk* h)
{
    StreamState& st = h->stream;
    if (st.launched) {
        const hipError_t q = hipEventQuery(st.evExit);
        if (q == hipErrorNotReady) return 0;
        if (q != hipSuccess) {
            g_err = std::string("stream: the resident launch failed: ") + hipGetErrorString(q);
            return 1;
        }
        if (int rc = stream_measure_launch(h)) return rc;     // it left by itself (idle time-out): book it
    }
    return stream_launch(h);
}

static int stream_free(psoap_chunk* h)
{
    StreamState& st = h->stream;
    (void)hipFree(st.dTasks); (void)hipFree(st.dLanes); (void)hipFree(st.dDev); (void)hipFree(st.dWs);
    (void)hipFree(st.dDag); (void)hipFree(st.dMats); (void)hipFree(st.dTlog);
    (void)hipHostFree(st.hHost); (void)hipHostFree(st.hLw); (void)hipHostFree(st.hGp);
    if (st.evExit) (void)hipEventDestroy(st.evExit);
    if (st.evStart) (void)hipEventDestroy(st.evStart);
    st = StreamState();
    return 0;
}

static int stream_open_impl(psoap_chunk* h, int c, int lanes, int scheme)
{
    StreamState& st = h->stream;
    st.C = c;
    st.lanes = lanes;
    const std::vector<int> all((size_t)lanes, h->P);
    if (scheme < 0) {
        const char* e = getenv("PSOAP_STREAM_SCHEME");
        scheme = e ? atoi(e) : dag_auto_scheme(all);
    }
    // every lane runs the task list of ONE matrix, cut as if `lanes` matrices shared the workers (one workgroup dispatches)
    // (PSOAP_STREAM_BURSTS=0: the lanes ticket by ticket in turn instead of a block row at a time -- experiments)
    const char* eb = getenv("PSOAP_STREAM_BURSTS");
    DagPlan plan = dag_build_lane_plan(h->P, lanes, h->dag_grid - 1, scheme, !(eb && eb[0] == '0'));
    st.scheme = plan.scheme;
    st.queues = plan.queues;
    st.n_tasks = (unsigned int)plan.tasks.size();
    st.ctrs_per_lane = plan.n_ctrs + 4;
    st.slots_per_lane = plan.n_slots + 1;
    if (const char* e = getenv("PSOAP_STREAM_IDLE_MS")) st.idle_ms = atof(e) > 0.0 ? atof(e) : st.idle_ms;
    HIP_TRY(hipMalloc(&st.dTasks, sizeof(DagTask) * plan.tasks.size()));
    HIP_TRY(hipMemcpy(st.dTasks, plan.tasks.data(), sizeof(DagTask) * plan.tasks.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc(&st.dLanes, sizeof(StreamLane) * lanes));
    {
        // next >= n_tasks: nothing to hand out.  (Not 0xffffffff: a worker's failed fetch-add would wrap it to 0.)
        std::vector<StreamLane> init((size_t)lanes);
        memset(init.data(), 0, sizeof(StreamLane) * lanes);
        for (StreamLane& ln : init) ln.next = 0x40000000u;
        HIP_TRY(hipMemcpy(st.dLanes, init.data(), sizeof(StreamLane) * lanes, hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMalloc(&st.dDev, sizeof(StreamDev)));
    {
        StreamDev init;
        memset(&init, 0, sizeof init);
        for (int x = 0; x < DAG_QUEUES; ++x) init.cur[x].lane = (unsigned int)(x % lanes);   // XCD x starts at its first lane
        HIP_TRY(hipMemcpy(st.dDev, &init, sizeof init, hipMemcpyHostToDevice));
    }
    HIP_TRY(hipHostMalloc(&st.hHost, sizeof(StreamHost), hipHostMallocCoherent));
    memset(st.hHost, 0, sizeof(StreamHost));
    st.h_stride = (size_t)c * h->N > 16 ? (size_t)c * h->N : 16;      // (orbital parameters: up to 13 doubles)
    HIP_TRY(hipHostMalloc(&st.hLw, sizeof(double) * (size_t)lanes * st.h_stride, hipHostMallocCoherent));
    HIP_TRY(hipHostMalloc(&st.hGp, sizeof(double) * (size_t)lanes * 2 * c, hipHostMallocCoherent));
    HIP_TRY(hipMalloc(&st.dWs, sizeof(double) * NB * NB * (size_t)st.slots_per_lane * lanes));
    st.arrive_off = sizeof(DagCtl) + sizeof(MatFlags) * (size_t)lanes;
    const size_t dag_bytes = st.arrive_off + sizeof(int) * (size_t)st.ctrs_per_lane * lanes;
    HIP_TRY(hipMalloc(&st.dDag, dag_bytes));
    HIP_TRY(hipMemset(st.dDag, 0, dag_bytes));
    HIP_TRY(hipMalloc(&st.dMats, sizeof(DagMat) * lanes));
    std::vector<DagMat> mats((size_t)lanes);
    for (int b = 0; b < lanes; ++b) {
        DagMat m{};
        m.K = h->dK + (size_t)b * h->mat_stride;
        m.R = h->dR + (size_t)b * h->Npad;
        m.Wt = h->dWt + (size_t)b * WT_STRIDE;
        m.lw = h->slot[0].dLwl + (size_t)b * c * h->N;       // the lane's device copy of its proposal (the dispatcher fills it)
        m.gp = h->slot[0].dGp + (size_t)b * 2 * c;
        m.sigma = h->dSigma;
        m.acc = h->dAcc + (size_t)b * ACC_ROWS;
        m.N = h->N;
        m.Npad = h->Npad;
        m.P = h->P;
        m.ld = h->ld;
        mats[b] = m;
    }
    HIP_TRY(hipMemcpy(st.dMats, mats.data(), sizeof(DagMat) * lanes, hipMemcpyHostToDevice));
    HIP_TRY(hipEventCreate(&st.evExit));
    HIP_TRY(hipEventCreate(&st.evStart));
    st.lane_ticket.assign((size_t)lanes, -1);
    st.lane_seq.assign((size_t)lanes, 0ull);
    st.lane_tries.assign((size_t)lanes, 0);
    st.lane_kind.assign((size_t)lanes, 0);
    st.lane_mu.assign((size_t)lanes, 1.0);
    st.neg.assign(STREAM_RING, 0);
    st.head = 0;
    st.open = true;
    return 0;
}

extern "C" int psoap_stream_open(psoap_chunk* h, int c, int lanes, int scheme)
{
    if (!h) FAIL("psoap_stream_open: null handle");
    if (c < 1 || c > 3) FAIL("psoap_stream_open: number of components must be 1, 2 or 3");
    if (lanes < 1 || lanes > h->max_batch || lanes > STREAM_MAX_LANES)
        FAIL("psoap_stream_open: 1 <= lanes <= min(max_batch, 64)");
    if (scheme < -1 || scheme > 2) FAIL("psoap_stream_open: scheme must be -1 (automatic), 0, 1 or 2");
    if (h->P > 255) FAIL("psoap_stream_open: N too large for the persistent kernel (N <= 32640)");
    if (h->stream.open) FAIL("psoap_stream_open: the handle already has an open stream");
    // (Late round 5 refused the following scheme (2) here: through a resident launch it returned a wrong value once in 20,000
    // ... 150,000 matrices.  Round 6 found the cause -- an unordered read-modify-write chain on the per-matrix accumulator
    // record and two progress words published out of order, both in the second level of following, neither specific to
    // streams -- and removed it (common.hpp MatAcc; dag_kernel.hpp dag_spin_ge; LABNOTES 14): all three schemes run here.)
    // the dispatcher keeps 3 x n_epochs velocities, 16 parameters and one flag per lane in the tile engine's LDS array
    if (h->n_epochs > 0 && (3 * (size_t)h->n_epochs + 16) * sizeof(double) + sizeof(int) * (size_t)lanes > GEMM_LDS_BYTES)
        FAIL("psoap_stream_open: too many epochs for the dispatcher's staging (3 n_epochs + 16 doubles + one int per lane must "
             "fit the tile engine's 73728 bytes of LDS)");
    if (share_wants_staged(h->device))
        FAIL("psoap_stream_open: too many processes share this GPU for a resident launch (more than PSOAP_SHARE_DAG_MAX, or "
             "several without the device lock): use the batch calls");
    DEVICE_SCOPE(h->device);
    if (int rc = enter_device(h->device)) return rc;
    if (int rc = psoap_chunk_sync(h)) return rc;         // batch evaluations of this handle use the same workspaces
    // the lanes' proposal arrays are those of proposal slot 0: whatever batch was uploaded is gone
    h->pend = -1;
    h->act = -1;
    h->slot[0].B = h->slot[1].B = 0;
    if (int rc = stream_open_impl(h, c, lanes, scheme)) {
        const std::string keep = g_err;
        (void)stream_free(h);
        g_err = keep;
        return rc;
    }
    return 0;
}

// kind: STREAM_IN_LWL (payload (n, c, N) ln-wavelengths), STREAM_IN_VELOCITIES ((n, c, n_epochs)), STREAM_IN_ORBITS ((n, np))
static int stream_submit_impl(psoap_chunk* h, int n, int kind, int model, const double* payload, size_t per, const double* gp,
                              double mu_GP, long long* tickets, const char* who)
{
    StreamState& st = h->stream;
    if (!st.open) { g_err = std::string(who) + ": no open stream (psoap_stream_open)"; return 2; }
    DEVICE_SCOPE(h->device);
    if (set_dev(h)) return 1;
    if (st.hHost->error != 0u) { g_err = std::string(who) + ": the stream has failed (a dependency wait timed out); close it"; return 2; }
    int free_lanes = 0;
    for (long long t : st.lane_ticket) free_lanes += (t < 0);
    if (n > free_lanes) { g_err = std::string(who) + ": not enough free lanes (fetch outstanding results first)"; return 2; }
    if (per > st.h_stride) { g_err = std::string(who) + ": the payload of one submission exceeds the lane's pinned buffer"; return 2; }
    // a result waits in the ring entry of its submission number: an unfetched one must not be lapped
    for (size_t l = 0; l < st.lane_ticket.size(); ++l)
        if (st.lane_ticket[l] >= 0 && st.lane_seq[l] + STREAM_RING <= st.head + (unsigned long long)n + (unsigned long long)st.lanes) {
            g_err = std::string(who) + ": a result submitted more than 192 submissions ago has not been fetched (fetch it first)";
            return 2;
        }
    // the device is this process's while the stream has tickets outstanding (taken BEFORE the head moves: a resident
    // dispatcher starts on the proposals at once); the fetch that takes the last result gives it back
    if (int rc = handle_lock(h)) return rc;
    const int c = st.C;
    int lane = 0;
    for (int k = 0; k < n; ++k) {
        while (st.lane_ticket[lane] >= 0) ++lane;
        const unsigned long long seq = st.head + (unsigned long long)k;
        memcpy(st.hLw + (size_t)lane * st.h_stride, payload + (size_t)k * per, sizeof(double) * per);
        memcpy(st.hGp + (size_t)lane * 2 * c, gp + (size_t)k * 2 * c, sizeof(double) * 2 * c);
        char neg = 0;
        for (int i = 0; i < 2 * c; ++i)
            if (gp[(size_t)k * 2 * c + i] < 0.0) neg = 1;      // covariance.py:317,339,362
        st.neg[seq % STREAM_RING] = neg;
        StreamEntry& e = st.hHost->entry[seq % STREAM_RING];
        e.lane = lane;
        e.kind = kind | (model << 8);
        e.mu = mu_GP;
        st.hHost->result[seq % STREAM_RING].seq1 = 0ull;
        st.lane_ticket[lane] = (long long)seq;
        st.lane_seq[lane] = seq;
        st.lane_tries[lane] = 0;
        st.lane_kind[lane] = e.kind;
        st.lane_mu[lane] = mu_GP;
        tickets[k] = (long long)seq;
    }
    // publish: everything above is visible before the new head (the device reads head, then the entries and proposals)
    __atomic_thread_fence(__ATOMIC_RELEASE);
    st.head += (unsigned long long)n;
    __atomic_store_n(&st.hHost->head, st.head, __ATOMIC_RELEASE);
    return stream_ensure_running(h);
}

extern "C" int psoap_stream_submit(psoap_chunk* h, int n, const double* lwl, const double* gp, double mu_GP,
                                   long long* tickets)
{
    if (!h || !lwl || !gp || !tickets || n < 1) FAIL("psoap_stream_submit: bad arguments");
    return stream_submit_impl(h, n, STREAM_IN_LWL, 0, lwl, (size_t)h->stream.C * h->N, gp, mu_GP, tickets, "psoap_stream_submit");
}

// radial velocities (n, c, n_epochs) instead of ln-wavelengths: the dispatcher shifts the chunk's grid (psoap_chunk_set_grid)
extern "C" int psoap_stream_submit_velocities(psoap_chunk* h, int n, const double* vel, const double* gp, double mu_GP,
                                              long long* tickets)
{
    if (!h || !vel || !gp || !tickets || n < 1) FAIL("psoap_stream_submit_velocities: bad arguments");
    if (!h->dGrid) FAIL("psoap_stream_submit_velocities: call psoap_chunk_set_grid first (before psoap_stream_open)");
    return stream_submit_impl(h, n, STREAM_IN_VELOCITIES, 0, vel, (size_t)h->stream.C * h->n_epochs, gp, mu_GP, tickets,
                              "psoap_stream_submit_velocities");
}

// orbital parameters (n, orbit_n_params(model)): Kepler solve, |v| >= c rule and Doppler shift in the dispatcher
extern "C" int psoap_stream_submit_orbits(psoap_chunk* h, int n, int model, const double* p_orb, const double* gp, double mu_GP,
                                          long long* tickets)
{
    if (!h || !p_orb || !gp || !tickets || n < 1) FAIL("psoap_stream_submit_orbits: bad arguments");
    if (!h->dGrid || !h->dDates)
        FAIL("psoap_stream_submit_orbits: call psoap_chunk_set_grid and psoap_chunk_set_dates first (before psoap_stream_open)");
    if (int rc = check_orbits(model, n, p_orb)) return rc;
    if (h->stream.open && orbit_n_components(model) != h->stream.C)
        FAIL("psoap_stream_submit_orbits: the model's number of components differs from the stream's");
    return stream_submit_impl(h, n, STREAM_IN_ORBITS, model, p_orb, (size_t)orbit_n_params(model), gp, mu_GP, tickets,
                              "psoap_stream_submit_orbits");
}

static int stream_pause_locked(psoap_chunk* h);

static int stream_lane_of(const StreamState& st, long long ticket)
{
    for (int l = 0; l < st.lanes; ++l)
        if (st.lane_ticket[l] == ticket) return l;
    return -1;
}

// Is the result of the matrix in `lane` there?  *ready = 1 and *lnp on yes.  A result whose matrix had a task on a
// workgroup that MOVED between compute units (StreamResult::tainted; dag_where in dag_kernel.hpp) is not handed out: the
// lane is submitted again -- its proposal still sits in the lane's pinned buffer -- under a new submission number, the
// caller's ticket stays what it was.  A stream has no staged path to fall back to: after 4 x PSOAP_SHARE_RETRIES (at least
// 4) resubmissions of one ticket the call fails loudly.
static int stream_lane_result(psoap_chunk* h, int lane, double* lnp, int* ready)
{
    StreamState& st = h->stream;
    const unsigned long long seq = st.lane_seq[lane];
    const size_t idx = (size_t)(seq % STREAM_RING);
    *ready = 0;
    if (__atomic_load_n(&st.hHost->result[idx].seq1, __ATOMIC_ACQUIRE) != seq + 1ull) return 0;
    bool tainted = __atomic_load_n(&st.hHost->result[idx].tainted, __ATOMIC_RELAXED) != 0ull;
    if (share_inject_taint()) tainted = true;
    if (!tainted) {
        if (lnp) *lnp = st.neg[idx] ? -INFINITY : st.hHost->result[idx].lnp;
        *ready = 1;
        return 0;
    }
    g_share.tainted += 1;
    const int max_tries = 4 * (share_retries() > 0 ? share_retries() : 1);
    if (st.lane_tries[lane] >= max_tries)
        FAIL("psoap_stream: a matrix was disturbed by the device's scheduler in every one of its resubmissions (too many processes "
             "share this GPU for a resident launch: use batch calls, or fewer processes)");
    ++st.lane_tries[lane];
    g_share.stream_resubmits += 1;
    const unsigned long long nseq = st.head;
    const size_t nidx = (size_t)(nseq % STREAM_RING);
    st.neg[nidx] = st.neg[idx];
    StreamEntry& e = st.hHost->entry[nidx];
    e.lane = lane;
    e.kind = st.lane_kind[lane];
    e.mu = st.lane_mu[lane];
    st.hHost->result[nidx].seq1 = 0ull;
    st.hHost->result[nidx].tainted = 0ull;
    st.lane_seq[lane] = nseq;
    __atomic_thread_fence(__ATOMIC_RELEASE);
    st.head += 1ull;
    __atomic_store_n(&st.hHost->head, st.head, __ATOMIC_RELEASE);
    return stream_ensure_running(h);
}

// 1: the result of `ticket` is there, 0: not yet (never blocks)
extern "C" int psoap_stream_ready(psoap_chunk* h, long long ticket, int* ready)
{
    if (!h || !ready || !h->stream.open) FAIL("psoap_stream_ready: bad arguments / no open stream");
    StreamState& st = h->stream;
    if (ticket < 0 || (unsigned long long)ticket >= st.head) FAIL("psoap_stream_ready: unknown ticket");
    const int lane = stream_lane_of(st, ticket);
    if (lane < 0) FAIL("psoap_stream_ready: the ticket was fetched before (or is too old)");
    if (set_dev(h)) return 1;
    if (int rc = stream_lane_result(h, lane, nullptr, ready)) return rc;
    // (a caller that only ever polls must still get the launch back after an idle exit)
    if (!*ready) return stream_ensure_running(h);
    return 0;
}

// A host-side bound on every wait for a result (seconds; PSOAP_STREAM_FETCH_TIMEOUT_S, default 120): the device's waits are
// bounded and reported, but a result that never comes for any other reason must end in an error, not in a hung process.
static double stream_fetch_timeout_s()
{
    const char* e = getenv("PSOAP_STREAM_FETCH_TIMEOUT_S");
    return (e && atof(e) > 0.0) ? atof(e) : 120.0;
}

static int stream_error_message(const StreamState& st, const char* who)
{
    char buf[320];
    snprintf(buf, sizeof buf,
             "%s: a dependency wait in the resident kernel timed out (results invalid); first failing wait: code=%u target=%u "
             "seen=%u", who, st.hHost->err_code, st.hHost->err_target, st.hHost->err_seen);
    FAIL(buf);
}

// blocks until ONE of the n tickets has its result; *which = its index in `tickets` (the lowest ready one)
extern "C" int psoap_stream_wait_any(psoap_chunk* h, int n, const long long* tickets, int* which)
{
    if (!h || !tickets || !which || n < 1) FAIL("psoap_stream_wait_any: bad arguments");
    StreamState& st = h->stream;
    if (!st.open) FAIL("psoap_stream_wait_any: no open stream");
    DEVICE_SCOPE(h->device);
    if (set_dev(h)) return 1;
    std::vector<int> lanes((size_t)n);
    for (int k = 0; k < n; ++k) {
        if (tickets[k] < 0 || (unsigned long long)tickets[k] >= st.head) FAIL("psoap_stream_wait_any: unknown ticket");
        lanes[k] = stream_lane_of(st, tickets[k]);
        if (lanes[k] < 0) FAIL("psoap_stream_wait_any: a ticket was fetched before (or is too old)");
    }
    long long spins = 0;
    const auto t_begin = std::chrono::steady_clock::now();
    const double limit = stream_fetch_timeout_s();
    for (;;) {
        for (int k = 0; k < n; ++k) {
            int ready = 0;
            if (int rc = stream_lane_result(h, lanes[k], nullptr, &ready)) return rc;
            if (ready) {
                *which = k;
                return 0;
            }
        }
        if ((++spins & 255) == 0) {
            if (__atomic_load_n(&st.hHost->error, __ATOMIC_ACQUIRE) != 0u) return stream_error_message(st, "psoap_stream_wait_any");
            if (int rc = stream_ensure_running(h)) return rc;
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count() > limit)
                FAIL("psoap_stream_wait_any: no result within PSOAP_STREAM_FETCH_TIMEOUT_S");
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
}

extern "C" int psoap_stream_fetch(psoap_chunk* h, int n, const long long* tickets, double* out)
{
    if (!h || !tickets || !out || n < 1) FAIL("psoap_stream_fetch: bad arguments");
    StreamState& st = h->stream;
    if (!st.open) FAIL("psoap_stream_fetch: no open stream");
    DEVICE_SCOPE(h->device);
    if (set_dev(h)) return 1;
    for (int k = 0; k < n; ++k) {
        const long long t = tickets[k];
        if (t < 0 || (unsigned long long)t >= st.head) FAIL("psoap_stream_fetch: unknown ticket");
        const int lane = stream_lane_of(st, t);
        if (lane < 0) FAIL("psoap_stream_fetch: the ticket was fetched before (or is too old)");
        long long spins = 0;
        const auto t_begin = std::chrono::steady_clock::now();
        const double limit = stream_fetch_timeout_s();
        for (;;) {
            int ready = 0;
            if (int rc = stream_lane_result(h, lane, &out[k], &ready)) return rc;
            if (ready) break;
            if ((++spins & 1023) == 0) {
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count() > limit)
                    FAIL("psoap_stream_fetch: no result within PSOAP_STREAM_FETCH_TIMEOUT_S (the stream is unusable: close it)");
                if (__atomic_load_n(&st.hHost->error, __ATOMIC_ACQUIRE) != 0u) return stream_error_message(st, "psoap_stream_fetch");
                // the launch may have ended (idle time-out) between this ticket's submission and its opening
                if (int rc = stream_ensure_running(h)) return rc;
            }
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        }
        st.lane_ticket[lane] = -1;
    }
    // Nothing outstanding any more: the device goes back to the other processes.  When there ARE others the resident
    // launch must have LEFT before the lock is given up -- it would otherwise sit on every compute unit for its idle
    // time-out while another process starts its own persistent launch (the first failure of DESIGN.md 5).
    bool outstanding = false;
    for (long long t : st.lane_ticket) outstanding = outstanding || t >= 0;
    if (!outstanding && h->dev_locked && st.launched && share_procs(h->device) > 1)
        if (int rc = stream_pause_locked(h)) return rc;
    handle_unlock(h);                     // (only if nothing is outstanding any more)
    return 0;
}

extern "C" int psoap_stream_stats(psoap_chunk* h, long long* launches, long long* submitted, long long* completed,
                                  int* scheme, long long* tasks_per_matrix)
{
    if (!h || !h->stream.open) FAIL("psoap_stream_stats: no open stream");
    DEVICE_SCOPE(h->device);
    if (set_dev(h)) return 1;
    const StreamState& st = h->stream;
    if (launches) *launches = st.launches;
    if (submitted) *submitted = (long long)st.head;
    if (completed) {
        long long done = 0;
        for (unsigned long long t = st.head > STREAM_RING ? st.head - STREAM_RING : 0; t < st.head; ++t)
            done += st.hHost->result[t % STREAM_RING].seq1 == t + 1ull;
        *completed = (st.head > STREAM_RING ? (long long)(st.head - STREAM_RING) : 0) + done;
    }
    if (scheme) *scheme = st.scheme;
    if (tasks_per_matrix) *tasks_per_matrix = (long long)st.n_tasks;
    return 0;
}

// Debug: per-task timestamps of the last `cap` submissions (8 x 100 MHz stamps per task, rows of n_tasks per
// submission, submission s in row s mod cap).  First call (out == NULL) allocates -- before the first submit.
extern "C" int psoap_stream_tasklog(psoap_chunk* h, int cap, unsigned long long* out, long long max_words)
{
    if (!h || !h->stream.open || cap < 1) FAIL("psoap_stream_tasklog: bad arguments / no open stream");
    StreamState& st = h->stream;
    DEVICE_SCOPE(h->device);
    if (set_dev(h)) return 1;
    const size_t words = (size_t)cap * st.n_tasks * 8;
    if (!out) {
        if (st.launched) FAIL("psoap_stream_tasklog: allocate before the first submission");
        if (st.dTlog) HIP_TRY(hipFree(st.dTlog));
        st.dTlog = nullptr;
        HIP_TRY(hipMalloc(&st.dTlog, sizeof(unsigned long long) * words));
        HIP_TRY(hipMemset(st.dTlog, 0, sizeof(unsigned long long) * words));
        st.tlog_cap = (unsigned int)cap;
        return 0;
    }
    if (!st.dTlog || (unsigned int)cap != st.tlog_cap) FAIL("psoap_stream_tasklog: no log of that capacity");
    // (the copy waits for the resident launch to leave: call it with nothing in flight)
    HIP_TRY(hipStreamSynchronize(h->streams[0]));
    const size_t n = (size_t)max_words < words ? (size_t)max_words : words;
    HIP_TRY(hipMemcpy(out, st.dTlog, sizeof(unsigned long long) * n, hipMemcpyDeviceToHost));
    return 0;
}

// Debug: the task list every lane runs (16-byte DagTask records, ticket order)
extern "C" int psoap_stream_tasks(psoap_chunk* h, void* out, long long max_tasks, long long* n_tasks)
{
    if (!h || !h->stream.open || !n_tasks) FAIL("psoap_stream_tasks: bad arguments / no open stream");
    DEVICE_SCOPE(h->device);
    if (set_dev(h)) return 1;
    const StreamState& st = h->stream;
    *n_tasks = st.n_tasks;
    if (out) {
        HIP_TRY(hipStreamSynchronize(h->streams[0]));
        const long long n = max_tasks < (long long)st.n_tasks ? max_tasks : (long long)st.n_tasks;
        HIP_TRY(hipMemcpy(out, st.dTasks, sizeof(DagTask) * n, hipMemcpyDeviceToHost));
    }
    return 0;
}

// (the launch has ended: its duration by the events around it, its matrices by the device's completion counter)
static int stream_measure_launch(psoap_chunk* h)
{
    StreamState& st = h->stream;
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, st.evStart, st.evExit));
    unsigned long long done = 0;
    HIP_TRY(hipMemcpy(&done, reinterpret_cast<char*>(st.dDev) + offsetof(StreamDev, completed), sizeof done,
                      hipMemcpyDeviceToHost));
    st.last_launch_ms = ms;
    st.last_launch_matrices = (long long)(done - st.completed_before);
    st.completed_before = done;
    st.launched = false;            // the next submit launches without asking the event
    return 0;
}

// The resident launch leaves NOW (once what is in flight is done) instead of after the idle time-out, and the call
// returns when it has: the device is free for other work (another handle's launch, a device-wide synchronise).  The
// stream stays open; the next submit brings the launch back.
extern "C" int psoap_stream_pause(psoap_chunk* h)
{
    if (!h) FAIL("psoap_stream_pause: null handle");
    StreamState& st = h->stream;
    if (!st.open) FAIL("psoap_stream_pause: no open stream");
    DEVICE_SCOPE(h->device);
    if (set_dev(h)) return 1;
    return stream_pause_locked(h);
}

static int stream_pause_locked(psoap_chunk* h)
{
    StreamState& st = h->stream;
    if (!st.launched) return 0;
    __atomic_store_n(&st.hHost->close, 1u, __ATOMIC_RELEASE);
    const hipError_t e = hipStreamSynchronize(h->streams[0]);
    __atomic_store_n(&st.hHost->close, 0u, __ATOMIC_RELEASE);
    if (e != hipSuccess) {
        g_err = std::string("psoap_stream_pause: ") + hipGetErrorString(e);
        return 1;
    }
    return stream_measure_launch(h);
}

// duration of the resident launch that ended last and the matrices it completed (valid after psoap_stream_pause)
extern "C" int psoap_stream_last_launch(psoap_chunk* h, double* ms, long long* matrices)
{
    if (!h || !h->stream.open) FAIL("psoap_stream_last_launch: no open stream");
    if (ms) *ms = h->stream.last_launch_ms;
    if (matrices) *matrices = h->stream.last_launch_matrices;
    return 0;
}

extern "C" int psoap_stream_close(psoap_chunk* h)
{
    if (!h) FAIL("psoap_stream_close: null handle");
    StreamState& st = h->stream;
    if (!st.open) return 0;
    DEVICE_SCOPE(h->device);
    if (set_dev(h)) return 1;
    int rc = 0;
    // what is in flight completes; what was published but never opened needs a launch to be consumed
    bool pending = false;
    for (size_t l = 0; l < st.lane_ticket.size(); ++l)
        if (st.lane_ticket[l] >= 0 &&
            __atomic_load_n(&st.hHost->result[st.lane_seq[l] % STREAM_RING].seq1, __ATOMIC_ACQUIRE) != st.lane_seq[l] + 1ull)
            pending = true;
    __atomic_store_n(&st.hHost->close, 1u, __ATOMIC_RELEASE);
    if (pending && st.hHost->error == 0u) rc = stream_ensure_running(h);
    if (hipStreamSynchronize(h->streams[0]) != hipSuccess) rc = rc ? rc : 1;
    (void)stream_free(h);
    handle_unlock(h, true);               // (the handle's stream is idle: nothing of this handle is in flight)
    return rc;
}

static int group_eval_locked(psoap_group* g, bool promote);

// The evaluation in flight has completed on the handle's stream: was it clean?  A persistent launch in which a workgroup
// MOVED between compute units while a task ran (DagCtl::pad[3], see dag_where) may have read a stale tile: the same launch
// is issued again -- same task list, so a clean run returns the bits an undisturbed one would have -- up to
// PSOAP_SHARE_RETRIES times, then the batch goes down the staged path (kernel boundaries only).  Never silently wrong.
static int settle_evaluation(psoap_chunk* h)
{
    for (int attempt = 0;; ++attempt) {
        HIP_TRY(hipStreamSynchronize(h->streams[0]));
        if (h->last_path != 1) return 0;
        if (h->hDagErr[0] != 0) return 0;                 // a timed-out wait: reported by the caller
        unsigned int moved = h->hDagErr[4];
        const unsigned int moved_xcd = h->hDagErr[5];
        if (share_inject_taint()) moved = moved ? moved : 1u;
        if (moved == 0u) return 0;
        g_share.tainted += 1;
        g_share.moved_tasks += (long long)moved;
        g_share.moved_xcd += (long long)moved_xcd;
        h->hDagErr[4] = h->hDagErr[5] = 0;
        if (share_detect_only()) return 0;
        if (attempt < share_retries()) {
            g_share.retries += 1;
            bool group_alive = false;
            if (h->last_group) {
                std::lock_guard<std::mutex> lk(g_groups_mu);
                group_alive = g_live_groups.count(h->last_group) != 0;
            }
            if (group_alive) {
                if (int rc = group_eval_locked(h->last_group, false)) return rc;
            } else if (int rc = eval_dag(h)) {
                return rc;
            }
            continue;
        }
        g_share.staged_fallbacks += 1;
        h->last_group = nullptr;          // (the other members of a group launch settle for themselves: same flags)
        if (int rc = eval_staged(h)) return rc;
    }
}

extern "C" int psoap_batch_fetch(psoap_chunk* h, double* out)
{
    if (!h || !out || h->act < 0 || h->slot[h->act].B < 1) FAIL("psoap_batch_fetch: nothing evaluated");
    DEVICE_SCOPE(h->device);
    if (set_dev(h)) return 1;
    const BatchSlot& sl = h->slot[h->act];
    const int rc_settle = settle_evaluation(h);
    // The device goes back to the other processes here, whatever is queued for this handle's NEXT evaluation: with an
    // upload pending (the pipelined loops) its copies are waited for first when the device is shared -- they are small, and
    // a collective or another process's launch must never find this process holding the lock (ranks that share a GPU
    // would wait for each other: one in the gather, one in flock)
    if (h->pend >= 0 && share_procs(h->device) > 1) (void)hipStreamSynchronize(h->copy);
    handle_unlock(h);
    if (rc_settle) return rc_settle;
    if (collect_timings(h)) return 1;
    if (h->last_path == 1 && h->hDagErr[0] != 0) {
        char buf[512];
        int dbg[32] = {0};
        (void)hipMemcpy(dbg, h->dDag, 16 * sizeof(int), hipMemcpyDeviceToHost);
        (void)hipMemcpy(dbg + 16, h->dDag + sizeof(DagCtl), 16 * sizeof(int), hipMemcpyDeviceToHost);
        snprintf(buf, sizeof buf,
                 "psoap_batch_fetch: a dependency wait in the DAG kernel timed out (results invalid); "
                 "first failing wait: code=%u target=%u seen=%u; err=%d; matrix0 rows_done=%d "
                 "potrf_done=%d cnt=%d",
                 h->hDagErr[1], h->hDagErr[2], h->hDagErr[3], dbg[1], dbg[16], dbg[17], dbg[18]);
        h->hDagErr[0] = 0;
        FAIL(buf);
    }
    for (int b = 0; b < sl.B; ++b) out[b] = sl.neg[b] ? -INFINITY : h->hOut[b];
    return 0;
}

extern "C" int psoap_chunk_sync(psoap_chunk* h)
{
    if (!h) FAIL("null handle");
    DEVICE_SCOPE(h->device);
    if (set_dev(h)) return 1;
    hipError_t e = hipSuccess;
    for (int g = 0; g < MAX_GROUPS; ++g)
        if (h->streams[g] && e == hipSuccess) e = hipStreamSynchronize(h->streams[g]);
    if (e == hipSuccess) e = hipStreamSynchronize(h->copy);
    handle_unlock(h);
    HIP_TRY(e);
    return 0;
}

extern "C" int psoap_chunk_get_timings(psoap_chunk* h, psoap_timings* t)
{
    if (!h || !t) FAIL("bad arguments");
    *t = h->last;
    return 0;
}

extern "C" int psoap_lnlike_batch(psoap_chunk* h, int B, int c, const double* lwl, const double* gp, double mu_GP,
                                  double* out)
{
    if (int rc = psoap_batch_upload(h, B, c, lwl, gp, mu_GP)) return rc;
    bool all_neg = true;
    for (int b = 0; b < B; ++b) all_neg = all_neg && h->slot[h->pend].neg[b];
    if (all_neg) {  // covariance.py:317-318: -inf before any work
        for (int b = 0; b < B; ++b) out[b] = -INFINITY;
        (void)hipStreamSynchronize(h->copy);      // (the upload's copies; nothing else was queued)
        handle_unlock(h);
        return 0;
    }
    if (int rc = psoap_batch_eval(h)) return rc;
    return psoap_batch_fetch(h, out);
}

extern "C" int psoap_lnlike(psoap_chunk* h, int c, const double* lwl, const double* gp, double mu_GP, double* out)
{
    return psoap_lnlike_batch(h, 1, c, lwl, gp, mu_GP, out);
}

// ---- fills (matrix_functions drop-ins) ------------------------------------------------------
extern "C" int psoap_fill_sym(int device, int c, int N, const double* lwl, const double* gp, const double* sigma,
                              double* out)
{
    if (c < 1 || c > 3 || N <= 0 || !lwl || !gp || !out) FAIL("psoap_fill_sym: bad arguments");
    DEVICE_SCOPE(device);
    HIP_TRY(hipSetDevice(device));
    const int Npad = round_up(N, NB), P = Npad / NB;
    DevBuf<double> dK, dLwl, dGp, dSig;
    HIP_TRY(dK.alloc((size_t)Npad * Npad));
    HIP_TRY(dLwl.alloc((size_t)c * N));
    HIP_TRY(dGp.alloc(6));
    HIP_TRY(hipMemcpy(dLwl, lwl, sizeof(double) * (size_t)c * N, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dGp, gp, sizeof(double) * 2 * c, hipMemcpyHostToDevice));
    if (sigma) {
        HIP_TRY(dSig.alloc(N));
        HIP_TRY(hipMemcpy(dSig, sigma, sizeof(double) * N, hipMemcpyHostToDevice));
    }
    dim3 grid(P * P, 1);
    if (c == 1)
        hipLaunchKernelGGL(k_fill_sym<1>, grid, dim3(256), 0, 0, dK.p, (size_t)0, Npad, N, P, dLwl.p, dGp.p, dSig.p, 0);
    else if (c == 2)
        hipLaunchKernelGGL(k_fill_sym<2>, grid, dim3(256), 0, 0, dK.p, (size_t)0, Npad, N, P, dLwl.p, dGp.p, dSig.p, 0);
    else
        hipLaunchKernelGGL(k_fill_sym<3>, grid, dim3(256), 0, 0, dK.p, (size_t)0, Npad, N, P, dLwl.p, dGp.p, dSig.p, 0);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy2D(out, sizeof(double) * N, dK, sizeof(double) * Npad, sizeof(double) * N, N,
                        hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int psoap_fill_cross(int device, int M, int N, const double* lwl_row, const double* lwl_col, double amp,
                                double l, double* out)
{
    if (M <= 0 || N <= 0 || !lwl_row || !lwl_col || !out) FAIL("psoap_fill_cross: bad arguments");
    DEVICE_SCOPE(device);
    HIP_TRY(hipSetDevice(device));
    const int ld = round_up(N, 2);
    DevBuf<double> dO, dRow, dCol;
    HIP_TRY(dO.alloc((size_t)M * ld));
    HIP_TRY(dRow.alloc(M));
    HIP_TRY(dCol.alloc(N));
    HIP_TRY(hipMemcpy(dRow, lwl_row, sizeof(double) * M, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dCol, lwl_col, sizeof(double) * N, hipMemcpyHostToDevice));
    dim3 grid((N + NB - 1) / NB, (M + NB - 1) / NB);
    hipLaunchKernelGGL(k_fill_cross, grid, dim3(256), 0, 0, dO.p, ld, M, N, dRow.p, dCol.p, amp, l);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy2D(out, sizeof(double) * N, dO, sizeof(double) * ld, sizeof(double) * N, M,
                        hipMemcpyDeviceToHost));
    return 0;
}

// ---- predict ---------------------------------------------------------------------------------
static int predict_check(int mode, int c, int N, int M, const void* lwl, const void* lwl_pred, const void* mu_c,
                         const void* gp, const void* mu_out)
{
    if (mode < 0 || mode > 2 || c < 1 || c > 3 || N <= 0 || M <= 0 || !lwl || !lwl_pred || !mu_c || !gp || !mu_out)
        FAIL("psoap_predict: bad arguments");
    if (mode == 2 && c != 1) FAIL("psoap_predict: mode 2 (predict_f) needs c == 1");
    return 0;
}

// predict_run until its persistent launch came through undisturbed (return code 3: a workgroup moved, see dag_where), then
// -- or at once where share_wants_staged says so -- by the staged loop of the transposed-mean variant
static int predict_settled(PredictWs& ws, int device, int mode, int c, int N, int M, const double* lwl, const double* fl,
                           const double* sigma, const double* dFl, const double* dSig, const double* lwl_pred,
                           const double* mu_c, const double* gp, double* mu_out, double* Sigma_out, int* status,
                           double* var_out)
{
    bool staged = share_wants_staged(device);
    if (staged) g_share.staged_policy += 1;
    for (int attempt = 0;; ++attempt) {
        if (!staged) g_share.dag_launches += 1;
        int rc = predict_run(ws, mode, c, N, M, lwl, fl, sigma, dFl, dSig, lwl_pred, mu_c, gp, mu_out, Sigma_out, status, g_err,
                             var_out, staged);
        if (rc == 0 && !staged && share_inject_taint()) rc = 3;
        if (rc != 3) return rc;
        g_share.tainted += 1;
        if (attempt < share_retries()) {
            g_share.retries += 1;
        } else {
            g_share.staged_fallbacks += 1;
            staged = true;
        }
    }
}

// handle-less form: a workspace for this one call (freed on every path by its destructor)
extern "C" int psoap_predict(int device, int mode, int c, int N, int M, const double* lwl, const double* fl,
                             const double* sigma, const double* lwl_pred, const double* mu_c, const double* gp,
                             double* mu_out, double* Sigma_out, int* status_out)
{
    if (!fl || !sigma) FAIL("psoap_predict: bad arguments");
    if (int rc = predict_check(mode, c, N, M, lwl, lwl_pred, mu_c, gp, mu_out)) return rc;
    DEVICE_SCOPE(device);
    if (int rc = enter_device(device)) return rc;
    PredictWs ws;
    if (int rc = dag_workers(device, &ws.workers, &ws.n_cus)) return rc;
    if (ws.workers > 2 * ws.n_cus) ws.workers = 2 * ws.n_cus;
    int status = 0;
    const int rc = predict_settled(ws, device, mode, c, N, M, lwl, fl, sigma, nullptr, nullptr, lwl_pred, mu_c, gp, mu_out,
                                   Sigma_out, &status, nullptr);
    (void)hipDeviceSynchronize();
    if (status_out) *status_out = status;
    return rc;
}

// explicit reusable workspace: the retrieve loop (scripts/psoap_retrieve_ST3.py:148) predicts once per chunk,
// chunk after chunk; one predictor serves them all and allocates only when a shape outgrows it
struct psoap_predictor {
    int device = 0;
    PredictWs ws;
};

static void copy_times(const PredictTimes& pt, psoap_predict_timings* t)
{
    t->device_ms = pt.device_ms;
    t->factor_ms = pt.factor_ms;
    t->sigma_ms = pt.sigma_ms;
    t->download_ms = pt.download_ms;
    t->total_ms = pt.total_ms;
    t->flops = pt.flops;
}

extern "C" int psoap_predictor_create(psoap_predictor** out, int device)
{
    if (!out) FAIL("psoap_predictor_create: bad arguments");
    *out = nullptr;
    DEVICE_SCOPE(device);
    if (int rc = enter_device(device)) return rc;
    psoap_predictor* p = new psoap_predictor();
    p->device = device;
    if (int rc = dag_workers(device, &p->ws.workers, &p->ws.n_cus)) {
        delete p;
        return rc;
    }
    if (p->ws.workers > 2 * p->ws.n_cus) p->ws.workers = 2 * p->ws.n_cus;
    *out = p;
    return 0;
}

extern "C" int psoap_predictor_destroy(psoap_predictor* p)
{
    if (!p) return 0;
    DEVICE_SCOPE_DESTROY(p->device);
    (void)hipSetDevice(p->device);
    (void)hipDeviceSynchronize();
    delete p;
    return 0;
}

extern "C" int psoap_predictor_run(psoap_predictor* p, int mode, int c, int N, int M, const double* lwl,
                                   const double* fl, const double* sigma, const double* lwl_pred, const double* mu_c,
                                   const double* gp, double* mu_out, double* Sigma_out, int* status_out)
{
    if (!p || !fl || !sigma) FAIL("psoap_predictor_run: bad arguments");
    if (int rc = predict_check(mode, c, N, M, lwl, lwl_pred, mu_c, gp, mu_out)) return rc;
    DEVICE_SCOPE(p->device);
    if (int rc = enter_device(p->device)) return rc;
    int status = 0;
    const int rc = predict_settled(p->ws, p->device, mode, c, N, M, lwl, fl, sigma, nullptr, nullptr, lwl_pred, mu_c, gp,
                                   mu_out, Sigma_out, &status, nullptr);
    if (status_out) *status_out = status;
    return rc;
}

// mean and diag(Sigma) only (see psoap_chunk_predict_var)
extern "C" int psoap_predictor_run_var(psoap_predictor* p, int mode, int c, int N, int M, const double* lwl,
                                       const double* fl, const double* sigma, const double* lwl_pred, const double* mu_c,
                                       const double* gp, double* mu_out, double* var_out, int* status_out)
{
    if (!p || !fl || !sigma || !var_out) FAIL("psoap_predictor_run_var: bad arguments");
    if (int rc = predict_check(mode, c, N, M, lwl, lwl_pred, mu_c, gp, mu_out)) return rc;
    DEVICE_SCOPE(p->device);
    if (int rc = enter_device(p->device)) return rc;
    int status = 0;
    const int rc = predict_settled(p->ws, p->device, mode, c, N, M, lwl, fl, sigma, nullptr, nullptr, lwl_pred, mu_c, gp,
                                   mu_out, nullptr, &status, var_out);
    if (status_out) *status_out = status;
    return rc;
}

extern "C" int psoap_predictor_timings(psoap_predictor* p, psoap_predict_timings* t)
{
    if (!p || !t) FAIL("psoap_predictor_timings: bad arguments");
    copy_times(p->ws.times, t);
    return 0;
}

// handle-resident form: fl / sigma are the handle's, every device buffer lives in a
// grow-only workspace owned by the handle, so a second call of the same shape allocates nothing.
static int chunk_predict(psoap_chunk* h, int mode, int c, int M, const double* lwl, const double* lwl_pred,
                         const double* mu_c, const double* gp, double* mu_out, double* Sigma_out, double* var_out,
                         int* status_out);

extern "C" int psoap_chunk_predict(psoap_chunk* h, int mode, int c, int M, const double* lwl, const double* lwl_pred,
                                   const double* mu_c, const double* gp, double* mu_out, double* Sigma_out,
                                   int* status_out)
{
    return chunk_predict(h, mode, c, M, lwl, lwl_pred, mu_c, gp, mu_out, Sigma_out, nullptr, status_out);
}

// mean and diag(Sigma) only: what the retrieve scripts plot (sqrt(diag(Sigma)), psoap_retrieve_ST3.py:111)
extern "C" int psoap_chunk_predict_var(psoap_chunk* h, int mode, int c, int M, const double* lwl, const double* lwl_pred,
                                       const double* mu_c, const double* gp, double* mu_out, double* var_out,
                                       int* status_out)
{
    if (!var_out) FAIL("psoap_chunk_predict_var: var_out is required");
    return chunk_predict(h, mode, c, M, lwl, lwl_pred, mu_c, gp, mu_out, nullptr, var_out, status_out);
}

static int chunk_predict(psoap_chunk* h, int mode, int c, int M, const double* lwl, const double* lwl_pred,
                         const double* mu_c, const double* gp, double* mu_out, double* Sigma_out, double* var_out,
                         int* status_out)
{
    if (!h) FAIL("psoap_chunk_predict: null handle");
    if (int rc = predict_check(mode, c, h->N, M, lwl, lwl_pred, mu_c, gp, mu_out)) return rc;
    DEVICE_SCOPE(h->device);
    if (int rc = enter_device(h->device)) return rc;
    if (!h->pws) {
        h->pws = new PredictWs();
        h->pws->workers = h->dag_grid < 2 * h->n_cus ? h->dag_grid : 2 * h->n_cus;       // (the AUG kernels: two per compute unit)
        h->pws->n_cus = h->n_cus;
    }
    int status = 0;
    const int rc = predict_settled(*h->pws, h->device, mode, c, h->N, M, lwl, nullptr, nullptr, h->dFl, h->dSigma, lwl_pred,
                                   mu_c, gp, mu_out, Sigma_out, &status, var_out);
    if (status_out) *status_out = status;
    return rc;
}

extern "C" int psoap_chunk_predict_timings(psoap_chunk* h, psoap_predict_timings* t)
{
    if (!h || !t) FAIL("psoap_chunk_predict_timings: bad arguments");
    if (!h->pws) FAIL("psoap_chunk_predict_timings: no predict call on this handle yet");
    copy_times(h->pws->times, t);
    return 0;
}

// drop the handle's predict workspace (it is also freed by psoap_chunk_destroy)
extern "C" int psoap_chunk_predict_release(psoap_chunk* h)
{
    if (!h) FAIL("psoap_chunk_predict_release: null handle");
    DEVICE_SCOPE(h->device);
    if (set_dev(h)) return 1;
    delete h->pws;
    h->pws = nullptr;
    return 0;
}

// ---- calibration ------------------------------------------------------------------------------
static int calibrate_check(int M, int N, int order, double lwl0, double lwl1)
{
    if (M <= 0 || N <= 0 || order < 0 || order > CAL_MAX_ORDER || !(lwl1 > lwl0)) return 1;
    return 0;
}

extern "C" int psoap_calibrate(int device, int c, int M, int N, int order, double lwl0, double lwl1,
                               const double* lwl_cal, const double* lwls_cal, const double* fl_cal,
                               const double* sigma_cal, const double* lwls_fixed, const double* fl_fixed,
                               const double* sigma_fixed, const double* gp, double mu_GP, double* fl_cor, double* X,
                               int* status_out)
{
    if (c < 1 || c > 3 || calibrate_check(M, N, order, lwl0, lwl1) || !lwl_cal || !lwls_cal || !fl_cal || !sigma_cal ||
        !lwls_fixed || !fl_fixed || !sigma_fixed || !gp || !fl_cor || !X)
        FAIL("psoap_calibrate: bad arguments");
    DEVICE_SCOPE(device);
    if (int rc = enter_device(device)) return rc;
    CalibInputs in{};
    in.M = M; in.N = N; in.order = order; in.lwl0 = lwl0; in.lwl1 = lwl1; in.mu = mu_GP;
    in.lwl_cal = lwl_cal; in.fl_cal = fl_cal; in.fl_fixed = fl_fixed;
    in.c = c; in.lwls_cal = lwls_cal; in.sigma_cal = sigma_cal; in.lwls_fixed = lwls_fixed; in.sigma_fixed = sigma_fixed;
    in.gp = gp;
    int status = 0;
    int rc = calibrate_run(in, fl_cor, X, &status, g_err);
    if (status_out) *status_out = status;
    return rc;
}

extern "C" int psoap_calibrate_explicit(int device, int M, int N, int order, double lwl0, double lwl1,
                                        const double* lwl_cal, const double* fl_cal, const double* fl_fixed,
                                        const double* A, const double* B, const double* C, double mu_GP,
                                        double* fl_cor, double* X, int* status_out)
{
    if (calibrate_check(M, N, order, lwl0, lwl1) || !lwl_cal || !fl_cal || !fl_fixed || !A || !B || !C || !fl_cor || !X)
        FAIL("psoap_calibrate_explicit: bad arguments");
    DEVICE_SCOPE(device);
    if (int rc = enter_device(device)) return rc;
    CalibInputs in{};
    in.M = M; in.N = N; in.order = order; in.lwl0 = lwl0; in.lwl1 = lwl1; in.mu = mu_GP;
    in.lwl_cal = lwl_cal; in.fl_cal = fl_cal; in.fl_fixed = fl_fixed;
    in.c = 0; in.A = A; in.B = B; in.C = C;
    int status = 0;
    int rc = calibrate_run(in, fl_cor, X, &status, g_err);
    if (status_out) *status_out = status;
    return rc;
}
