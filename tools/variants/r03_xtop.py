"""What-if: X rows only up to band 48 (the headline header's kx + m = 45, rounded up to 64 bytes): k_hfps does not
store the bands above, k_synth does not load them (zeros instead).  Invalid audio when anything non-zero lives there;
bounds what a `top`-aware X hand-over could gain.  name = xtop<bands>[s|l]: s = stores only, l = loads only."""
import re, sys
d, name = sys.argv[1], sys.argv[2]
m = re.match(r'xtop(\d+)([sl]?)', name)
top, what = int(m.group(1)), m.group(2)
if what != 'l':
    p = d + '/k_psf.h'; s = open(p).read()
    old = '''            X.stb(lv.x, qb, n * 64);          X.stb(lv.y, qb, XP + n * 64);
            X.stb(rr.x, qb, 2 * XP + n * 64); X.stb(rr.y, qb, 3 * XP + n * 64);'''
    assert old in s
    s = s.replace(old, 'if (q < %d) {\n' % top + old + '\n}')
    open(p, 'w').write(s)
if what != 's':
    p = d + '/k_he.hip'; s = open(p).read()
    old = '''        const f32x4 t = syn_ld4(q < 8 ? p0 + q * 64 + lane : p1 + (q - 8) * 64 + lane);'''
    assert old in s
    s = s.replace(old, '''        f32x4 t = {0.0f, 0.0f, 0.0f, 0.0f};
        if ((lane & 15) * 4 < %d) t = syn_ld4(q < 8 ? p0 + q * 64 + lane : p1 + (q - 8) * 64 + lane);''' % top)
    open(p, 'w').write(s)
