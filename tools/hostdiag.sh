echo "THP: $(cat /sys/kernel/mm/transparent_hugepage/enabled) defrag: $(cat /sys/kernel/mm/transparent_hugepage/defrag)"
echo "nodes: $(ls -d /sys/devices/system/node/node* | wc -l)"; for n in /sys/devices/system/node/node*; do echo "$n cpus $(cat $n/cpulist) mem $(grep MemTotal $n/meminfo | awk '{print $4}') kB"; done
for c in /sys/class/drm/card*/device/numa_node; do echo "$c $(cat $c)"; done
nproc; taskset -p $$ | head -2
cat /proc/self/status | grep -i "cpus_allowed_list\|mems_allowed_list"
