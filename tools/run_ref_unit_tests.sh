#!/bin/bash
# Run the staged reference unit tests against the drop-in, one interpreter per file, verbose (debugging aid for
# tests/test_reference_unit_tests_gpu.py): tools/run_ref_unit_tests.sh [file ...] -> gpurun_out/refunit_<file>.log
cd "$(dirname "$0")/.." || exit 1
ROOT=$(pwd)
mkdir -p gpurun_out
files=("$@")
[ ${#files[@]} -eq 0 ] && files=(test_layers.py test_mel_loss.py test_parallel_wavegan.py test_melgan.py test_hifigan.py test_style_melgan.py)
for f in "${files[@]}"; do
  (cd oracle/_ref/test && PYTHONPATH=$ROOT PYTHONFAULTHANDLER=1 AMD_LOG_LEVEL=0 timeout 900 python -m pytest -p tests.refunit.plugin -v \
     --no-header -p no:cacheprovider "$f" > "$ROOT/gpurun_out/refunit_$f.log" 2>&1)
  echo "== $f rc=$? : $(grep -c PASSED gpurun_out/refunit_$f.log) passed, $(grep -c FAILED gpurun_out/refunit_$f.log) failed"
  tail -3 "gpurun_out/refunit_$f.log"
done
