# Build container: a variant of the library for A/B runs on the GPU box.
#   bash tools/experiments/r03/build_variant.sh NAME "-DFOO=1 ..."   ->  dustraytracer_amd/ab_NAME.so   (git-ignored, travels with gpurun)
set -e
NAME=$1; EXTRA=$2
ROOT=$(cd $(dirname $0)/../../.. && pwd)
W=/tmp/drt_variant_$NAME; rm -rf $W; mkdir -p $W/dustraytracer_amd $W/include
cp -r $ROOT/dustraytracer_amd/csrc $W/dustraytracer_amd/; cp $ROOT/include/*.h* $W/include/
rm -f $W/dustraytracer_amd/csrc/*.o
make -C $W/dustraytracer_amd/csrc -j8 EXTRA="$EXTRA" > $W/build.log 2>&1 || { tail -20 $W/build.log; exit 1; }
cp $W/dustraytracer_amd/libdrt_hip.so $ROOT/dustraytracer_amd/ab_$NAME.so
echo "built dustraytracer_amd/ab_$NAME.so ($EXTRA)"
