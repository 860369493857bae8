// comm_shm.cpp -- see comm_shm.h.  Reference precedent for "the adapter owns its communicator": IpMumpsSolverInterface.cpp:58-75
// (MPI_Init inside the linear-solver interface).
#include "comm_shm.h"
#include <hip/hip_runtime_api.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <random>
#include <thread>
#include <vector>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace mi355x {

namespace {
constexpr uint32_t SHM_MAGIC = 0x4d48534bu;       // "KSHM"
constexpr int MAX_RANKS = 64;

struct alignas(64) Bar { std::atomic<uint32_t> count; std::atomic<uint32_t> gen; };
struct Header {
    std::atomic<uint32_t> magic;
    uint32_t nranks;
    uint64_t slot_bytes, slots_off, total_bytes;
    std::atomic<uint32_t> attached;
    std::atomic<uint32_t> abort_flag;                 // a rank that timed out raises it: the others stop waiting too
};
// layout: Header | Bar[nranks * (nranks + 1)] (one barrier per range [lo, lo + size)) | pad to 4 KiB | nranks slots
size_t bars_off() { return (sizeof(Header) + 63) / 64 * 64; }
size_t nbars(int P) { return (size_t)P * (size_t)(P + 1); }

double timeout_s()
{
    static const double t = getenv("MI355X_KKT_SHM_TIMEOUT_S") ? atof(getenv("MI355X_KKT_SHM_TIMEOUT_S")) : 300.0;
    return t > 0 ? t : 300.0;
}
size_t slot_bytes_wanted()
{
    static const double mib = getenv("MI355X_KKT_SHM_SLOT_MIB") ? atof(getenv("MI355X_KKT_SHM_SLOT_MIB")) : 8.0;
    size_t b = (size_t)((mib > 0 ? mib : 8.0) * 1048576.0);
    return (b + 4095) / 4096 * 4096;
}
struct Mapping { void* base = nullptr; size_t bytes = 0; std::string name; };
std::mutex g_mu;
std::map<std::string, Mapping> g_created;            // segments this process created (rank 0) and has not attached to yet
}  // namespace

struct ShmComm {
    Mapping m;
    int rank = 0, nranks = 1;
    std::vector<unsigned char> acc;                   // the sum of one chunk (host), copied back to the device
    Header* hdr() const { return reinterpret_cast<Header*>(m.base); }
    Bar* bar(int lo, int size) const { return reinterpret_cast<Bar*>((char*)m.base + bars_off()) + ((size_t)lo * (size_t)(nranks + 1) + (size_t)size); }
    unsigned char* slot(int r) const { return (unsigned char*)m.base + hdr()->slots_off + (size_t)r * hdr()->slot_bytes; }
    // sense-reversing barrier among the `size` ranks of [lo, lo + size); false on time-out / abort
    bool barrier(int lo, int size) const {
        if (size <= 1) return true;
        Bar* b = bar(lo, size);
        const uint32_t g = b->gen.load(std::memory_order_acquire);
        if (b->count.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)size) {
            b->count.store(0, std::memory_order_relaxed);
            b->gen.store(g + 1, std::memory_order_release);
            return true;
        }
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spin = 0; b->gen.load(std::memory_order_acquire) == g; ++spin) {
            if (hdr()->abort_flag.load(std::memory_order_relaxed)) return false;
            if (spin < 2000) sched_yield();
            else {
                std::this_thread::sleep_for(std::chrono::microseconds(50));
                if ((spin & 1023) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s()) {
                    hdr()->abort_flag.store(1, std::memory_order_relaxed);
                    return false;
                }
            }
        }
        return true;
    }
};

bool shm_comm_create(int nranks, void* out128, std::string& err)
{
    if (nranks < 1 || nranks > MAX_RANKS || !out128) { err = "shm communicator: 1 .. 64 ranks"; return false; }
    std::random_device rd;
    char name[128];
    snprintf(name, sizeof name, "/mi355x_kkt_%ld_%08x%08x", (long)getpid(), (unsigned)rd(), (unsigned)rd());
    const size_t slot = slot_bytes_wanted();
    const size_t slots_off = (bars_off() + nbars(nranks) * sizeof(Bar) + 4095) / 4096 * 4096;
    const size_t total = slots_off + (size_t)nranks * slot;
    const int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) { err = std::string("shm_open(create ") + name + "): " + strerror(errno); return false; }
    if (ftruncate(fd, (off_t)total) != 0) { err = std::string("ftruncate(shm): ") + strerror(errno); close(fd); shm_unlink(name); return false; }
    void* base = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (base == MAP_FAILED) { err = std::string("mmap(shm): ") + strerror(errno); shm_unlink(name); return false; }
    Header* h = new (base) Header();
    h->nranks = (uint32_t)nranks; h->slot_bytes = slot; h->slots_off = slots_off; h->total_bytes = total;
    h->attached.store(0); h->abort_flag.store(0);
    Bar* b = reinterpret_cast<Bar*>((char*)base + bars_off());
    for (size_t i = 0; i < nbars(nranks); ++i) { b[i].count.store(0); b[i].gen.store(0); }
    h->magic.store(SHM_MAGIC, std::memory_order_release);
    std::memset(out128, 0, 128);
    std::memcpy(out128, name, strlen(name) + 1);
    std::lock_guard<std::mutex> lk(g_mu);
    Mapping m; m.base = base; m.bytes = total; m.name = name;
    g_created[name] = m;
    return true;
}

ShmComm* shm_comm_attach(const void* id128, int rank, int nranks, std::string& err)
{
    if (!id128 || rank < 0 || rank >= nranks || nranks > MAX_RANKS) { err = "shm communicator: bad rank / size"; return nullptr; }
    char name[129];
    std::memcpy(name, id128, 128); name[128] = 0;
    if (name[0] != '/' || std::strncmp(name, "/mi355x_kkt_", 12) != 0) { err = "shm communicator: the 128-byte id is not one of shm_comm_create"; return nullptr; }
    Mapping m;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_created.find(name);
        if (it != g_created.end()) { m = it->second; g_created.erase(it); }
    }
    if (!m.base) {
        const int fd = shm_open(name, O_RDWR, 0600);
        if (fd < 0) { err = std::string("shm_open(") + name + "): " + strerror(errno); return nullptr; }
        struct stat sb;
        if (fstat(fd, &sb) != 0 || (size_t)sb.st_size < sizeof(Header)) { err = "shm communicator: segment too small"; close(fd); return nullptr; }
        m.bytes = (size_t)sb.st_size; m.name = name;
        m.base = mmap(nullptr, m.bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (m.base == MAP_FAILED) { err = std::string("mmap(shm): ") + strerror(errno); return nullptr; }
    }
    ShmComm* c = new ShmComm();
    c->m = m; c->rank = rank; c->nranks = nranks;
    Header* h = c->hdr();
    if (h->magic.load(std::memory_order_acquire) != SHM_MAGIC || (int)h->nranks != nranks || h->total_bytes != m.bytes) {
        err = "shm communicator: the segment does not describe this job (magic / rank count)";
        munmap(m.base, m.bytes); delete c; return nullptr;
    }
    c->acc.resize(h->slot_bytes);
    h->attached.fetch_add(1, std::memory_order_acq_rel);
    const auto t0 = std::chrono::steady_clock::now();
    while (h->attached.load(std::memory_order_acquire) < (uint32_t)nranks) {
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s() || h->abort_flag.load()) {
            h->abort_flag.store(1);
            err = "shm communicator: not every rank attached within MI355X_KKT_SHM_TIMEOUT_S";
            if (rank == 0) shm_unlink(name);
            munmap(m.base, m.bytes); delete c; return nullptr;
        }
    }
    if (rank == 0) shm_unlink(name);        // every rank holds its mapping: the name can go (no leak if the job dies later)
    return c;
}

void shm_comm_discard(const void* id128)
{
    if (!id128) return;
    char name[129];
    std::memcpy(name, id128, 128); name[128] = 0;
    if (std::strncmp(name, "/mi355x_kkt_", 12) != 0) return;
    Mapping m;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_created.find(name);
        if (it == g_created.end()) return;          // attached already (attach unlinks the name itself) or not ours
        m = it->second; g_created.erase(it);
    }
    shm_unlink(name);
    if (m.base) munmap(m.base, m.bytes);
}

void shm_comm_destroy(ShmComm* c)
{
    if (!c) return;
    if (c->m.base) munmap(c->m.base, c->m.bytes);
    delete c;
}

template <class T> static void sum_slots(const ShmComm* c, T* acc, size_t n, int lo, int size)
{
    const T* s0 = reinterpret_cast<const T*>(c->slot(lo));
    for (size_t i = 0; i < n; ++i) acc[i] = s0[i];
    for (int r = lo + 1; r < lo + size; ++r) {                       // rank order: every rank of the range forms bitwise the same sum
        const T* s = reinterpret_cast<const T*>(c->slot(r));
        for (size_t i = 0; i < n; ++i) acc[i] += s[i];
    }
}

int shm_comm_allreduce_range(void* ctx, void* dptr, int64_t count, int dtype, void* hip_stream, int lo, int size)
{
    ShmComm* c = static_cast<ShmComm*>(ctx);
    if (!c || count < 0 || lo < 0 || size < 1 || lo + size > c->nranks || c->rank < lo || c->rank >= lo + size) return 1;
    if (count == 0) return 0;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    const size_t esz = dtype == 0 ? 8 : 4;
    const size_t per = c->hdr()->slot_bytes / esz;
    if (hipStreamSynchronize(st) != hipSuccess) return 2;          // ordered after the work already enqueued on the stream
    for (int64_t off = 0; off < count; off += (int64_t)per) {
        const size_t n = (size_t)std::min<int64_t>((int64_t)per, count - off);
        char* d = static_cast<char*>(dptr) + (size_t)off * esz;
        if (hipMemcpyAsync(c->slot(c->rank), d, n * esz, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return 2;
        if (!c->barrier(lo, size)) return 3;
        if (dtype == 0) sum_slots<double>(c, reinterpret_cast<double*>(c->acc.data()), n, lo, size);
        else            sum_slots<int32_t>(c, reinterpret_cast<int32_t*>(c->acc.data()), n, lo, size);
        if (!c->barrier(lo, size)) return 3;                         // every rank has read every slot: they may be overwritten
        if (hipMemcpyAsync(d, c->acc.data(), n * esz, hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return 2;   // (on the solver's stream: ordered before whatever it enqueues next)
    }
    return 0;
}

int shm_comm_allreduce(void* ctx, void* dptr, int64_t count, int dtype, void* hip_stream)
{
    ShmComm* c = static_cast<ShmComm*>(ctx);
    return c ? shm_comm_allreduce_range(ctx, dptr, count, dtype, hip_stream, 0, c->nranks) : 1;
}

}  // namespace mi355x
