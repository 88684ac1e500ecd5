#!/bin/bash
# gpurun call that carries the variant libraries (cacophony_amd/_variants/, kept out of ordinary pushes by .gpurunignore):
#   bash tools/gpurun_variants.sh --timeout 1800 -- 'bash tools/gpu_session.sh variants'
cd "$(dirname "$0")/.."
cp .gpurunignore /tmp/.gpurunignore.saved
grep -v '^cacophony_amd/_variants/' /tmp/.gpurunignore.saved > .gpurunignore
/usr/local/graft/bin/gpurun "$@"; rc=$?
cp /tmp/.gpurunignore.saved .gpurunignore
exit $rc
