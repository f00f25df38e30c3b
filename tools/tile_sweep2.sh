for c in "16 128 128 3 1 76" "16 64 64 3 1 152" "16 64 64 1 1 304" "16 32 64 3 1 304" "16 128 128 1 1 76" "16 64 128 3 2 152" "16 256 128 1 1 76" "16 128 256 3 1 76"; do
  for t in 128x128 192x128 160x128 256x128 96x128 128x64 192x64 256x64 160x64; do
    CY_IGEMM_TILE=$t python tools/conv_micro.py $c 20 fwd,dgrad 2>/dev/null | sed "s/^/$t /"
  done
done
