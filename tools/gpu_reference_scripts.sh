#!/bin/bash
# Run the reference's own driver scripts, byte-unchanged, on an MI355X box (tests/test_reference_scripts_gpu.py).
# /root/reference does not exist there and its sources must not enter this repository's history: they are staged under the
# git-ignored oracle/_ref/scripts/ for the duration of ONE gpurun call and removed again, whatever the outcome.
#   tools/gpu_reference_scripts.sh [gpurun timeout in seconds] [pytest -k expression]   (from the repository root, in the
#   build container)
set -u
REF=${REFERENCE_ROOT:-/root/reference}
STAGE=oracle/_ref/scripts
OUT=gpurun_out/r03_refscripts
mkdir -p "$STAGE" "$OUT"
trap 'rm -rf "$STAGE"; rmdir oracle/_ref 2>/dev/null' EXIT
for f in convert_weight.py test_single_image.py video_test.py eval.py train.py args.py; do
    cp "$REF/$f" "$STAGE/$f" || exit 1
done
( cd "$STAGE" && sha256sum convert_weight.py test_single_image.py video_test.py eval.py train.py args.py ) > "$OUT/staged_sha256.txt"
( cd "$REF" && sha256sum convert_weight.py test_single_image.py video_test.py eval.py train.py args.py ) > "$OUT/reference_sha256.txt"
cmp "$OUT/staged_sha256.txt" "$OUT/reference_sha256.txt" || exit 1
/usr/local/graft/bin/gpurun --timeout "${1:-1800}" -- "mkdir -p $OUT; timeout ${1:-1800} python -m pytest tests/test_reference_scripts_gpu.py tests/test_compat.py -m gpu -x -q -s -rs ${2:+-k \"$2\"} > $OUT/tests.log 2>&1; echo rc=\$?; grep -E 'passed|failed|skipped|reference, compat|test_single_image.py:|video_test.py|Error' $OUT/tests.log | tail -12"
