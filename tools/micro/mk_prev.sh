#!/bin/bash
# build the COMMITTED tree's library as variant `prev` (csrc/_build/variants/prev/libjsorb.so), then the working tree's again: A/B of uncommitted kernel changes
cd $(dirname $0)/../..
git stash -q || exit 1
python -c "from jetson_slam_amd import build as b; b.build_lib()" && mkdir -p jetson_slam_amd/csrc/_build/variants/prev && cp jetson_slam_amd/libjsorb.so jetson_slam_amd/csrc/_build/variants/prev/libjsorb.so
git stash pop -q
python -c "from jetson_slam_amd import build as b; b.build_lib(); print('prev + current built')"
