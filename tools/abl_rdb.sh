set -e
for v in base nowload noxread nomma nowx stageonly; do
  echo "== $v"
  if [ $v = base ]; then unset SRX_LIB_PATH; else export SRX_LIB_PATH=$PWD/tools/variants/libsrx_$v.so; fi
  timeout -k 10 100 python tools/mb_rdb.py 16,64,64 16,32,32 64,128,128,3,3 2>&1 | grep -v Warn
done
