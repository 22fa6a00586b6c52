for v in "PWG_DBG=0" "PWG_DBG=16" "PWG_DBG=32" "PWG_DBG=64"; do
  echo "=== $v"
  env $v timeout 200 python tools/bench_conv.py 16 800 2>&1 | grep -v amdgpu.ids | awk '{print $1,$2,$3,$4, $(NF-3), $(NF-2), $(NF-1), $NF}'
done
