for shape in "128 65536" "256 65536" "64 262144" "32 1048576" "512 32768" "1024 16384"; do
  for mi in 4096 1; do
    LLZ_TUNE=iir_wave_min_items=$mi SEGS=0 python tools/iir_segs.py $shape 2>&1 | grep auto | sed "s/^/min_items=$mi /"
  done
done
