set -x
mkdir -p /tmp/w && cd /tmp/w
python - <<'PY'
import sys; sys.path.insert(0, '/root/repo')
from pathlib import Path
from tests.test_gpu_cli import write_dataset
write_dataset(Path('/tmp/w/data'), 300, 200, 9000, 17)
PY
timeout 120 /root/repo/hgaprec_amd/hgaprec -dir /tmp/w/data -n 300 -m 200 -k 5 -hier -rfreq 2 -max-iterations 4 -ngpus 2 -device 0 ; echo "exit=$?"
