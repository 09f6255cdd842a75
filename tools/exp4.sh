# all three configs: rocprofv3 kernel stats of the dominant kernels (tools/exp3.sh per workload)
for w in config2 config3 config5; do echo "#### $w"; bash tools/exp3.sh $w "$@"; done
