# GPU box: A/B of library builds over the benchmark scenes: bash tools/ab_scenes.sh libA.so libB.so ...
for wl in "cornell_box 1920 1080 8" "suzanne_plane 1920 1080 8" "dense_monkey 1920 1080 16" "cs16_dust 1920 1080 8" "room 1920 1080 8" "mc_transparency 1920 1080 8"; do
  echo "== $wl"
  python tools/ab_libs.py "$@" -- $wl
done
