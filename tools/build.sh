#!/bin/bash
# build librejit_hip.so; non-zero exit (and the compiler errors) when it fails
set -o pipefail
cd "$(dirname "$0")/.."
python -c "import rejit_amd; rejit_amd.build()" 2>&1 | grep -E "error|Error" -A6 | head -40
exit ${PIPESTATUS[0]}
