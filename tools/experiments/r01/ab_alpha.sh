for wl in "mc_transparency 1920 1080 8" "mc_transparency 843 460 50" "uv_texture_test 1920 1080 8"; do
  echo "== $wl"
  python tools/ab_libs.py "$@" -- $wl
done
