# same-box A/B of the library against yolopoint_amd/lib/ab/libOLD.so (the build before a change): training records
P="import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
for i in 1 2; do
  echo -n "new s64: "; python bench.py --mode train --batch 64 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "$P"
  echo -n "old s64: "; YP_HIP_LIB=yolopoint_amd/lib/ab/libOLD.so python bench.py --mode train --batch 64 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "$P"
  echo -n "new l16 fp8: "; python bench.py --mode train --version l --batch 16 --dtype fp8 --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "$P"
  echo -n "old l16 fp8: "; YP_HIP_LIB=yolopoint_amd/lib/ab/libOLD.so python bench.py --mode train --version l --batch 16 --dtype fp8 --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "$P"
done
