#!/bin/bash
# tools/build_variant.sh NAME "-DFLAG=..." : cross-compiles a tuning build of libwfpt.so into build/libwfpt_NAME.so
# (select it at run time with WFPT_LIB=build/libwfpt_NAME.so). The default library is wavefront_path_tracer_amd/libwfpt.so.
set -e
cd "$(dirname "$0")/.."
WFPT_EXTRA_FLAGS="$2" WFPT_LIB_OUT="$PWD/build/libwfpt_$1.so" python -m wavefront_path_tracer_amd._build >/dev/null
echo "built build/libwfpt_$1.so ($2)"
