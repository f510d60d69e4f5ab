#!/bin/bash
# what the GPU box offers for clock / power sampling and for a C client (probe; output to gpurun_out/probe_env.txt)
OUT=gpurun_out/probe_env.txt; mkdir -p gpurun_out
{
echo "== sysfs"; for c in /sys/class/drm/card*/device; do echo $c; ls $c | tr '\n' ' ' | head -c 1500; echo; for f in pp_dpm_sclk pp_dpm_mclk gpu_busy_percent current_link_speed; do [ -r $c/$f ] && { echo "-- $f"; head -12 $c/$f; }; done
  for h in $c/hwmon/hwmon*; do echo "-- $h"; ls $h | tr '\n' ' '; echo; for f in power1_average power1_input freq1_input freq2_input power1_cap temp1_input; do [ -r $h/$f ] && echo "$f=$(cat $h/$f 2>&1)"; done; done; done
echo "== python amdsmi"; python -c "import amdsmi; print('amdsmi ok', amdsmi.__file__)" 2>&1 | tail -1
echo "== which"; which amd-smi rocm-smi rocprofv3
echo "== time rocm-smi"; ( time rocm-smi --showclocks --showpower --json ) 2>&1 | tail -8
echo "== time amd-smi"; ( time amd-smi metric -g 0 --clock --power --json ) 2>&1 | tail -40
echo "== ldconfig hip"; ldconfig -p | grep -i amdhip; cat /etc/ld.so.conf.d/*rocm* 2>/dev/null
echo "== nproc"; nproc; lscpu | head -20
} > $OUT 2>&1
