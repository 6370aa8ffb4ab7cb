"""Variant: k_synth draws frames from its queue QCH at a time (one atomic per QCH frames instead of per frame).
usage: VARIANT_EDIT=tools/variants/r03_queue_chunk.py tools/build_variants.sh synq4 ""   (name = synq<QCH>)"""
import re, sys
d, name = sys.argv[1], sys.argv[2]
qch = int(re.match(r'synq(\d+)', name).group(1))
p = d + '/k_he.hip'
s = open(p).read()
def rep(a, b):
    global s
    assert a in s, a
    s = s.replace(a, b)
rep('''    // two tickets: the frame in work and the next one (first one: the wave's own index)
    unsigned long long f = (unsigned long long)blockIdx.x * NW + wave;
    unsigned tk = 0;
    if (lane == 0) tk = atomicAdd(g_queue, 1u) + gridDim.x * NW;
    unsigned long long f1 = (unsigned long long)__builtin_amdgcn_readfirstlane(tk);
    SynIn cur;
    if (f < n_frames) load_unit(f, 0, cur);
    while (f < n_frames) {
        unsigned nxt = 0;
        if (lane == 0) nxt = atomicAdd(g_queue, 1u) + gridDim.x * NW;''', '''    // frames are drawn QCH at a time: the chunk in work and the next one (first chunk: the wave's own index)
    constexpr unsigned QCH = %du;
    unsigned long long f = ((unsigned long long)blockIdx.x * NW + wave) * QCH;
    unsigned pos = 0;
    unsigned tk = 0;
    if (lane == 0) tk = atomicAdd(g_queue, QCH) + gridDim.x * NW * QCH;
    unsigned long long nextbase = (unsigned long long)__builtin_amdgcn_readfirstlane(tk);
    SynIn cur;
    if (f < n_frames) load_unit(f, 0, cur);
    while (f < n_frames) {
        const bool last = pos + 1 == QCH;
        const unsigned long long f1 = last ? nextbase : f + 1;
        unsigned nxt = 0;
        if (last && lane == 0) nxt = atomicAdd(g_queue, QCH) + gridDim.x * NW * QCH;''' % qch)
rep('''        f = f1;
        f1 = (unsigned long long)__builtin_amdgcn_readfirstlane(nxt);
    }
}

// Stage-level batched filterbanks''', '''        f = f1;
        if (last) { nextbase = (unsigned long long)__builtin_amdgcn_readfirstlane(nxt); pos = 0; }
        else pos++;
    }
}

// Stage-level batched filterbanks''')
open(p, 'w').write(s)
