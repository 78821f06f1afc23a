for shape in "512 512 2 5 10 10" "256 512 2 5 10 10" "256 256 2 10 20 20" "512 256 2 10 20 20" "128 256 2 10 20 20" "128 128 2 20 40 40" "256 128 2 20 40 40"; do
  for nt in 0 64; do for ks in 0 1 3 9 27; do
    FPLX_TILE_NT=$nt FPLX_TILE_KS=$ks python tools/conv_bench.py $shape 2>/dev/null | sed "s/rows=.*env=/ /"
  done; done
done
