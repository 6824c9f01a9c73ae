set -u
cd $GRAFT_REPO_ROOT
lib=raytracing_amd/csrc/librt_hip.so
cp $lib /tmp/keep.so
for v in raytracing_amd/csrc/variants/*.so; do
  cp $v $lib
  echo "== $(basename $v)"
  timeout -k 10 200 python tools/gpu_perf.py 1024 0 2>&1 | grep -v amdgpu.ids | cut -c1-190 || break
done
cp /tmp/keep.so $lib
