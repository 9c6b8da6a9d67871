"""Apply (or revert with `git checkout`) the cycle-stamp instrumentation of csrc/knn_rows_mfma.hip used by
tools/knn_phase_cycles.py, then build with -DFSG_KNN_STATS.  Not part of the product build."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
p = os.path.join(ROOT, "fissure-segmentation_amd", "csrc", "knn_rows_mfma.hip")
s = open(p).read()
def rep(old, new):
    global s
    assert old in s, old[:60]
    s = s.replace(old, new, 1)
rep("__device__ unsigned long long fsg_knn_stats[8];", "__device__ unsigned long long fsg_knn_stats[32];")
rep("unsigned long long z[8] = {0};", "unsigned long long z[32] = {0};")
rep('''#define KSTATMAX(i, v) atomicMax(&fsg_knn_stats[i], (unsigned long long)(v))
#else''', '''#define KSTATMAX(i, v) atomicMax(&fsg_knn_stats[i], (unsigned long long)(v))
#define KT0() unsigned long long kt_ = (wave == KW && lane == 0) ? __builtin_readcyclecounter() : 0ull
#define KTICK(i)                                                                  \\
    do {                                                                          \\
        if (wave == KW && lane == 0) {                                            \\
            const unsigned long long n_ = __builtin_readcyclecounter();           \\
            KSTAT(i, n_ - kt_);                                                   \\
            kt_ = n_;                                                             \\
        }                                                                         \\
    } while (0)
#else
#define KT0() ((void)0)
#define KTICK(i) ((void)0)''')
rep('''    if (QAL) __syncthreads();   // A operand copy complete
''', '''    if (QAL) __syncthreads();   // A operand copy complete
    const int KW = (flags >> 24) & 15;   // which wave is stamped
    KT0();
''')
rep('''                    }
                    }
                }
                __syncthreads();
''', '''                    }
                    }
                }
                KTICK(8);     // phase A of this wave
                __syncthreads();
                KTICK(9);     // waiting for the other waves' phase A
''')
rep('''                    const int cc = ccount[qi];
                    unsigned tau;''', '''                    const int cc = ccount[qi];
                    KTICK(10);    // row read
                    unsigned tau;''')
rep('''                    // compact the survivors behind the carried list: per-lane count, wave prefix sum, per-lane stores
                    if (lane < cc) sv[lane] = carry[qi * CK + lane];''', '''                    KTICK(11);    // threshold
                    // compact the survivors behind the carried list: per-lane count, wave prefix sum, per-lane stores
                    if (lane < cc) sv[lane] = carry[qi * CK + lane];''')
rep('''                    if (total <= SURV) {
                        rank_merge(sv, qi, cc, total);
                    } else {''', '''                    KTICK(12);    // count + scan + survivor stores
                    if (total <= SURV) {
                        rank_merge(sv, qi, cc, total);
                        KTICK(13);   // rank by counting
                        if (wave == KW && lane == 0) { KSTAT(16, total - cc); KSTAT(17, 1); }
                    } else {''')
rep('''                __syncthreads();  // rows are rewritten by the next chunk; carry/ccount visible
            }
            if (!redo || attempt == 1) break;''', '''                KTICK(14);    // tail of phase B
                __syncthreads();  // rows are rewritten by the next chunk; carry/ccount visible
                KTICK(15);    // waiting for the other waves' phase B
            }
            if (!redo || attempt == 1) break;''')
open(p, "w").write(s)
print("instrumented", p)
