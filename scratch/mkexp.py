"""Build scratch/libsrx_<NAME>.so: conv.hip with parts of the K loop removed (timing experiments only;
results are wrong).  usage: mkexp.py NAME [noread] [nobar] [nomma]"""
import subprocess, sys
name, flags = sys.argv[1], set(sys.argv[2:])
s = open('scratch/conv_xchunk_prod.hip.txt').read()   # the experimental kernel: cross-chunk read pipelining + optional producer waves
def rep(old, new):
    global s
    assert s.count(old) == 1, (s.count(old), old[:60])
    s = s.replace(old, new)
if 'noread' in flags:
    # fragments are read once (both buffers) before the loop and never again
    rep('''  auto read_row = [&](int stg, int dy, int buf) {''', '''  bool primed = false;
  auto read_row = [&](int stg, int dy, int buf) {
    if (primed) return;''')
    rep('''  if (p.nchunk > 0) open_chunk(0, 0, 0);''', '''  read_row(0, 0, 0); read_row(0, 0, 1); primed = true;
  if (p.nchunk > 0) open_chunk(0, 0, 0);''')
if 'nobar' in flags:
    s = s.replace('__builtin_amdgcn_s_barrier();', '')
if 'nomma' in flags:
    rep('''mma(wf[(P + dy) & 1][dx][n], xf[(P + dy) & 1][dx][rg], acc[rg][n]);''',
        '''asm volatile("" :: "v"(wf[(P + dy) & 1][dx][n]), "v"(xf[(P + dy) & 1][dx][rg]));''')
open('scratch/conv_exp_%s.hip' % name, 'w').write(s)
c = ("cd scratch && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../include -I../super-resolution_amd/csrc -c conv_exp_%s.hip -o conv_%s.o && "
     "hipcc -shared -fPIC --offload-arch=gfx950 ../super-resolution_amd/csrc/api.o conv_%s.o ../super-resolution_amd/csrc/conv_pair.o "
     "../super-resolution_amd/csrc/wgrad.o ../super-resolution_amd/csrc/elementwise.o ../super-resolution_amd/csrc/tiles.o -o libsrx_%s.so") % (name, name, name, name)
subprocess.check_call(c, shell=True)
print("built", name)
