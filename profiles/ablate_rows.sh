export WQ_REPS=40
for ab in 0 1 2 4 5 6; do
  for rows in 20 205 2049; do
    echo "ablate=$ab rows=$rows: $(TBK_ABLATE_GRID=$ab WQ_ROWS=$rows python profiles/wave_quantisation.py 2>&1 | grep 'rows ' | cut -c40-58)"
  done
done
./profiles/microbench/stream_write_nt 16 | grep tiny
