"""Timing variants of the weight-pack kernel (results wrong): noload | nostore | nolds"""
import subprocess, sys
name, flags = sys.argv[1], set(sys.argv[2:])
s = open('super-resolution_amd/csrc/elementwise.hip').read()
def rep(old, new):
    global s
    assert s.count(old) == 1, (s.count(old), old[:60])
    s = s.replace(old, new)
for fl in flags:
    if fl.startswith('lpb'):
        rep('constexpr int PACK_LPB = 8;', 'constexpr int PACK_LPB = %s;' % fl[3:])
if 'noload' in flags:
    rep("v[t] = src ? src[t] * sc : 0.f;", "v[t] = src ? sc + (float)t : 0.f;")
if 'nostore' in flags:
    rep("store_oct(d + (long)o * 8, &tile[t * PACK_PLANE + r * 8]);", "if (tile[t * PACK_PLANE + r * 8] == 123.456f) store_oct(d + (long)o * 8, &tile[t * PACK_PLANE + r * 8]);")
if 'nobias' in flags:
    rep("if (blockIdx.x == 0 && L.dst_bias) {", "if (false) {")
open('scratch/elementwise_%s.hip' % name, 'w').write(s)
c = ("cd scratch && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../include -I../super-resolution_amd/csrc -c elementwise_%s.hip -o ew_%s.o && "
     "hipcc -shared -fPIC --offload-arch=gfx950 ../super-resolution_amd/csrc/api.o ../super-resolution_amd/csrc/conv.o ../super-resolution_amd/csrc/conv_pair.o "
     "../super-resolution_amd/csrc/wgrad.o ew_%s.o ../super-resolution_amd/csrc/tiles.o -o libsrx_%s.so") % (name, name, name, name)
subprocess.check_call(c, shell=True)
print("built", name)
