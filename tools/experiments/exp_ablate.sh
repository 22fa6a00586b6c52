mkdir -p gpurun_out/r4k
cd /root/repo
export PWG_PAIR_D=0
for A in "" msd0 msd mpd "msd,mpd0,mpd1,mpd2,mpd3" "msd,mpd1,mpd2,mpd3,mpd4" "mpd,msd1,msd2" "mpd,msd0" mpd0 mpd4 "msd1,msd2" "mpd0,mpd1" nofm; do
  PWG_ABL="$A" python tools/ablate_c3.py c3 14 2>&1 | tail -1 >> gpurun_out/r4k/ablate.txt
done
cat gpurun_out/r4k/ablate.txt
