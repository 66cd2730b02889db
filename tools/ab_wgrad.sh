for g in "12 48 160 256 64 1" "12 48 160 64 256 1" "12 24 80 512 128 1" "12 12 40 1024 256 1" "12 6 20 2048 512 1" "12 48 160 64 64 3" "12 24 80 128 128 3" "12 12 40 256 256 3" "12 6 20 512 512 3"; do
  echo "== $g"
  for lib in "" "--lib tools/bin/libsqd_HEAD.so"; do
    python tools/bench_wgrad.py $lib $g 2>&1 | grep -E "direct fp32|shared fp32" | sort -t: -k2 -n | head -2 | sed "s#^#   [${lib:-new}] #"
  done
done
