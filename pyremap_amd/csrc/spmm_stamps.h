// spmm_stamps.h -- in-kernel time stamps (diagnostic build only).
// Part of remap_spmm.hip: included there inside namespace remap::(anonymous),
// in the order given there; not a stand-alone header.
// ---------------------------------------------------------------------------
// In-kernel stamps (diagnostic build only: -DREMAP_STAMPS, tools/stamps.sh).
// Each stamp reads s_memtime and drains the scalar counter (the recipe of
// cdna_hip_programming.md section 7); the VM flavour first drains vmcnt.  The
// five phase sums of a wave go to a buffer of their own (the launch passes
// it in KParams::mask_out, the byte mask being unused then); no output value
// depends on them.  The product build compiles all of this away.
// ---------------------------------------------------------------------------
#ifdef REMAP_STAMPS
#define REMAP_STAMP_INIT()                                                   \
    unsigned long long st_prev = 0, st_now = 0;                              \
    unsigned long long st_sum[5] = {0, 0, 0, 0, 0};                          \
    unsigned long long st_rows = 0
#define REMAP_STAMP(k)                                                       \
    do {                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                   \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)"                  \
                     : "=s"(st_now)::"memory");                              \
        __builtin_amdgcn_sched_barrier(0);                                   \
        if ((k) != 0)                                                        \
            st_sum[k] += st_now - st_prev;                                   \
        else                                                                 \
            st_rows += 1;                                                    \
        st_prev = st_now;                                                    \
    } while (0)
#define REMAP_STAMP_VM(k)                                                    \
    do {                                                                     \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     \
        REMAP_STAMP(k);                                                      \
    } while (0)
#define REMAP_STAMP_FLUSH()                                                  \
    do {                                                                     \
        if (lane == 0 && p.mask_out) {                                       \
            unsigned long long *o =                                          \
                reinterpret_cast<unsigned long long *>(p.mask_out);          \
            for (int k = 1; k < 5; ++k)                                      \
                atomicAdd(o + k, st_sum[k]);                                 \
            atomicAdd(o, st_rows);                                           \
        }                                                                    \
    } while (0)
#else
#define REMAP_STAMP_INIT() do { } while (0)
#define REMAP_STAMP(k) do { } while (0)
#define REMAP_STAMP_VM(k) do { } while (0)
#define REMAP_STAMP_FLUSH() do { } while (0)
#endif
