#!/bin/bash
# usage: tools/asan_host.sh   (CPU, build container): libtdthost.so rebuilt with AddressSanitizer + UBSan in a temporary copy of the
# package and the host-side tests run against it (GPU sanitizers are not available on the pool: the HIP side is covered by parity tests)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
cp -r "$ROOT" "$TMP/repo"
cd "$TMP/repo"
g++ -O1 -g -std=c++17 -fPIC -shared -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off -I include \
    tdt4230_project_raytracing_amd/csrc/host_scene.cpp tdt4230_project_raytracing_amd/csrc/host_view.cpp -o tdt4230_project_raytracing_amd/libtdthost.so -lz
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python -m pytest tests/test_host.py tests/test_ply_ingest.py tests/test_present.py \
    tests/test_camera_controller.py tests/test_octree_util.py -x -q
rm -rf "$TMP"
