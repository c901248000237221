O=gpurun_out/g4; mkdir -p $O
export MBR_PROBE_SPLIT=1
python tools/mbr_probe.py block_2 block_4 block_7 block_10 > $O/probe_base.txt 2>&1
python tools/_relink.py mbr.hip -DMBR_EXP_ONE_TAP > /dev/null 2>&1
python tools/mbr_probe.py block_2 block_4 block_7 block_10 > $O/probe_onetap.txt 2>&1
tail -12 $O/probe_base.txt; echo ----; tail -12 $O/probe_onetap.txt
