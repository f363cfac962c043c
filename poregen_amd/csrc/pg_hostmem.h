// pg_hostmem.h -- host-side storage of the library's big result arrays (pg_api.hip, pg_job.hip; not installed)
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <sys/mman.h>
#include <memory>
#include <new>
#include <vector>

// ---- big host arrays of the library (downloaded / merged kept samples: hundreds of MB at large limits) ----------------------
// vector<T> whose resize() leaves new elements uninitialised (they are overwritten by the download / merge right away, zero-filling
// them first costs as much as the copy) and whose storage, from 4 MB up, is a mapping of its own with transparent huge pages asked
// for: the runtime pins 2 MB pages instead of 4 KB pages when it copies into it (3-8 x faster for pageable memory, profiles/
// r03_e2e_hugepages.txt), the first touch takes 512 x fewer faults, and so does the release.
template <class T> struct NoInitAlloc {
    using value_type = T;
    NoInitAlloc() = default;
    template <class U> NoInitAlloc(const NoInitAlloc<U> &) {}
    static constexpr size_t kHugePage = (size_t)2 << 20;
    static size_t mapped(size_t n) { const size_t b = n * sizeof(T); return b >= 2 * kHugePage ? (b + kHugePage - 1) & ~(kHugePage - 1) : 0; }
    T *allocate(size_t n) {
        if (const size_t m = mapped(n)) {
            void *q = mmap(nullptr, m, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (q == MAP_FAILED) throw std::bad_alloc();
#ifdef MADV_HUGEPAGE
            (void)madvise(q, m, MADV_HUGEPAGE);
#endif
            return static_cast<T *>(q);
        }
        return static_cast<T *>(::operator new(n * sizeof(T)));
    }
    void deallocate(T *p, size_t n) { if (const size_t m = mapped(n)) munmap(p, m); else ::operator delete(p); }
    template <class U, class... A> void construct(U *p, A &&...a) {
        if constexpr (sizeof...(A) == 0) ::new ((void *)p) U; else ::new ((void *)p) U(std::forward<A>(a)...);
    }
    template <class U> bool operator==(const NoInitAlloc<U> &) const { return true; }
    template <class U> bool operator!=(const NoInitAlloc<U> &) const { return false; }
};
using SampleVec = std::vector<double, NoInitAlloc<double>>;
using BigVec32 = std::vector<uint32_t, NoInitAlloc<uint32_t>>;
using BigVec64 = std::vector<uint64_t, NoInitAlloc<uint64_t>>;
