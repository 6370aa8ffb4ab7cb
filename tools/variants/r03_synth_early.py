"""What-if: k_synth issues the next unit's loads right after the staging transposes (before the IMDCTs)
instead of after them.  usage: VARIANT_EDIT=tools/variants/r03_synth_early.py tools/build_variants.sh synearly "" """
import sys
d = sys.argv[1]
p = d + '/k_he.hip'
s = open(p).read()
def rep(a, b):
    global s
    assert a in s, a
    s = s.replace(a, b)
rep('''template <class SL>
__device__ __forceinline__ void syn_rows(const SL &S, SynWave &w, const SynIn &d, int lane)
{''', '''template <class SL, class Hook = NoHook>
__device__ __forceinline__ void syn_rows(const SL &S, SynWave &w, const SynIn &d, int lane, Hook after_stage = Hook())
{''')
rep('''            x[4 * q] = t.x; x[4 * q + 1] = t.y; x[4 * q + 2] = t.z; x[4 * q + 3] = t.w;
        }
        wave_sync();
    }
    SSTAMP(1);''', '''            x[4 * q] = t.x; x[4 * q + 1] = t.y; x[4 * q + 2] = t.z; x[4 * q + 3] = t.w;
        }
        wave_sync();
    }
    float hh[18];
#pragma unroll
    for (int r = 0; r < 18; r++) hh[r] = d.h[r];
    after_stage();
    SSTAMP(1);''')
rep('''        w.vb[(32 + (r >> 1)) * VB_STRIDE + (r & 1) * 64 + lane] = d.h[r];''', '''        w.vb[(32 + (r >> 1)) * VB_STRIDE + (r & 1) * 64 + lane] = hh[r];''')
rep('''            syn_rows(S, w, cur, lane);
            wave_sync();
            if (ch + 1 < nout) load_unit(f, ch + 1, cur);
            else if (f1 < n_frames) load_unit(f1, 0, cur);
            syn_poly<1>''', '''            SynIn nxtin;
            syn_rows(S, w, cur, lane, [&]() {
                if (ch + 1 < nout) load_unit(f, ch + 1, nxtin);
                else if (f1 < n_frames) load_unit(f1, 0, nxtin);
            });
            wave_sync();
            cur = nxtin;
            syn_poly<1>''')
open(p, 'w').write(s)
