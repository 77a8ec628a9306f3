#!/bin/bash
# Build the HIP library (and the oracle's C half) from anywhere.  usage: tools/build.sh [-B]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
make -C "$ROOT/hydrodem_amd/csrc" -j8 "$@" libhydrodem_hip.so 2>&1 | grep -E "error|warning|Error" 
make -C "$ROOT/oracle" -s liboracle_c.so
ls -la --time-style=+%H:%M:%S "$ROOT/hydrodem_amd/csrc/libhydrodem_hip.so" | awk '{print $6, $7}'
