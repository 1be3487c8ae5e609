set -e
for n in 1 16 1 16; do echo "== weight sets $n"; MB_WEIGHT_SETS=$n timeout -k 10 120 python tools/mb_shapes.py 16,64,64,64,32,208 16,64,64,160,32,208 16,64,64,192,64,208; done
