set -e
for p in 0 1; do
  for shape in "10000 3" "100000 16" "1000000 16" "500 3" "1000000 4"; do
    set -- $shape
    MHLIK_CHAIN_PERSIST=$p timeout -k 10 200 python bench.py --cpu-iters 0 --no-extras --rows $1 --cols $2 --steps 8000 --warmup 6000
  done
done
