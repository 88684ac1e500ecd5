#!/bin/bash
# gpurun call that carries the variant libraries (cacophony_amd/_variants/, kept out of ordinary pushes by .gpurunignore):
#   bash tools/gpurun_variants.sh --timeout 1800 -- 'bash tools/gpu_session.sh variants'
# The tracked .gpurunignore is restored on every way out (normal exit, Ctrl-C, a timeout's TERM): a lifted line left behind would make
# every later ordinary push carry the variant libraries.
cd "$(dirname "$0")/.."
ls cacophony_amd/_variants/*.so >/dev/null 2>&1 || { echo "no variant libraries: run tools/build_variants.sh first"; exit 1; }
SAVED=$(mktemp /tmp/gpurunignore.XXXXXX)
cp .gpurunignore "$SAVED"
trap 'cp "$SAVED" .gpurunignore; rm -f "$SAVED"' EXIT
trap 'exit 130' INT TERM
grep -v '^cacophony_amd/_variants/' "$SAVED" > .gpurunignore
/usr/local/graft/bin/gpurun "$@"
