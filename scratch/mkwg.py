"""Timing variants of wgrad.hip (results wrong): nodma | noread | nobar | nomma"""
import subprocess, sys
name, flags = sys.argv[1], set(sys.argv[2:])
s = open('super-resolution_amd/csrc/wgrad.hip').read()
def rep(old, new, cnt=1):
    global s
    assert s.count(old) == cnt, (s.count(old), old[:60])
    s = s.replace(old, new)
if 'nodma' in flags:      # only the first tile is staged
    rep("    if (more) stage_tile(tile + 1, stage ^ 1);", "")
if 'oobdma' in flags:     # every staging load out of range: issue cost without memory traffic
    rep('ok ? xoffs[i] : OOB', 'OOB')
    rep('ok ? goffs[i] : OOB', 'OOB')
if 'nobar' in flags:
    rep('''    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    stage ^= 1;''', '''    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stage ^= 1;''')
if 'noread' in flags:     # fragments come from the first tile's registers only
    rep("for (; tile < t_end; ++tile) {", "bool first = true; frag_t keepx[3] = {}; frag_t keepg = {};\n  for (; tile < t_end; ++tile) {")
    rep("gf[y][c] = load_g(sgb, y * NC + c);", "{ if (first) keepg = load_g(sgb, y * NC + c); gf[y][c] = keepg; }")
    rep("for (int dx = 0; dx < 3; ++dx) xf[dx] = load_x(sxb, ry, c * 16 + dx);", "for (int dx = 0; dx < 3; ++dx) { if (first) keepx[dx] = load_x(sxb, ry, c * 16 + dx); xf[dx] = keepx[dx]; }")
    rep('''    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    stage ^= 1;''', '''    first = false;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    stage ^= 1;''') if 'nobar' not in flags else rep("    stage ^= 1;\n  }\n\n  if (!active) return;", "    first = false;\n    stage ^= 1;\n  }\n\n  if (!active) return;")
open('scratch/wgrad_%s.hip' % name, 'w').write(s)
c = ("cd scratch && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -I../include -I../super-resolution_amd/csrc -c wgrad_%s.hip -o wg_%s.o && "
     "hipcc -shared -fPIC --offload-arch=gfx950 ../super-resolution_amd/csrc/api.o ../super-resolution_amd/csrc/conv.o ../super-resolution_amd/csrc/conv_pair.o "
     "wg_%s.o ../super-resolution_amd/csrc/elementwise.o ../super-resolution_amd/csrc/tiles.o -o libsrx_%s.so") % (name, name, name, name)
subprocess.check_call(c, shell=True)
print("built", name)
