#!/bin/bash
# Build the committed HEAD of the library into build_ab/lib_old.so for A/B timing against the working tree.
set -e
cd "$(dirname "$0")/.."
rm -rf /tmp/oldsrc && git worktree prune && git worktree add -f /tmp/oldsrc HEAD >/dev/null 2>&1
make -C /tmp/oldsrc/ccnet_amd/csrc -j8 2>&1 | grep -E "error" || true
mkdir -p build_ab && cp /tmp/oldsrc/ccnet_amd/libccnet_hip.so build_ab/lib_old.so
git worktree remove --force /tmp/oldsrc
