cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=/root/repo
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fe -o fe -- python3 /root/repo/tools/r03_frontend_ab.py 10 > /tmp/fe.log 2>&1
grep "ms$" /tmp/fe.log
f=$(find /tmp/fe -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -12 "$f" | cut -c1-150
