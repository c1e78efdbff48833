cd /root/repo
PYTHONPATH=/root/repo timeout 300 python tools/r03_frontend_ab.py 10 2>&1 | tail -8
