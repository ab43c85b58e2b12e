export DRT_TW_SUN=1
for wl in "sunshadow_test 1920 1080 8" "mc_transparency 1920 1080 8" "cornell_box 1920 1080 8" "cs16_dust 1920 1080 4" "lightweight_rt 1920 1080 8"; do
  echo "== sun: $wl"
  python tools/ab_libs.py "$@" -- $wl
done
unset DRT_TW_SUN
for wl in "cornell_box 1920 1080 8" "cs16_dust 1920 1080 8"; do
  echo "== no sun: $wl"
  python tools/ab_libs.py "$@" -- $wl
done
