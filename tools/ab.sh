#!/bin/bash
# A/B two builds of libpq_amd.so on ONE GPU box: the working tree (alt) against HEAD (base).
# usage: tools/ab.sh "<kbench args>"   -- run from the repo root, on the build machine
set -e
cp vllm_pq_amd/libpq_amd.so /tmp/libpq_worktree.so
git stash -q
make -C vllm_pq_amd/csrc -j8 2>&1 | grep -E "error" || true
cp vllm_pq_amd/libpq_amd.so vllm_pq_amd/libpq_base.so
git stash pop -q
cp /tmp/libpq_worktree.so vllm_pq_amd/libpq_amd.so
touch vllm_pq_amd/libpq_amd.so
