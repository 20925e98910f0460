#!/bin/bash
# Round 5, verdict item 1: is real OpenCV reachable from the GPU box?  Three probes, each bounded; if any finds cv2 the
# fixture recipe runs and its outputs come home in gpurun_out/.  Log: gpurun_out/cv2_probe.log
mkdir -p gpurun_out
L=gpurun_out/cv2_probe.log
: > $L
{
echo "== probe 1: import cv2 (every python on the box)"
for py in python python3 /usr/bin/python3 /opt/conda/bin/python /opt/conda/bin/python3.9 $(ls /opt/conda/envs/*/bin/python 2>/dev/null); do
  [ -x "$(command -v $py)" ] || continue
  echo "-- $py"; timeout 120 $py -c "import cv2; print('cv2', cv2.__version__, cv2.__file__)" 2>&1 | tail -1
done
echo "== probe 2: find cv2 / libopencv on the filesystem"
timeout 120 find / -xdev \( -name 'cv2*' -o -name 'libopencv*' -o -name 'opencv*.whl' -o -name 'opencv_python*' \) -not -path '/proc/*' 2>/dev/null | head -20
echo "== probe 3: pip install --target /tmp/cv opencv-python-headless (is there a network?)"
timeout 90 python -m pip install --no-input --disable-pip-version-check --target /tmp/cv opencv-python-headless 2>&1 | tail -4
echo "-- pip config / indexes"; python -m pip config list 2>&1 | head; env | grep -i -E 'pip_|proxy' | head
echo "-- DNS / route"; timeout 10 getent hosts pypi.org; timeout 10 python - <<'PY'
import socket
for h, p in (("pypi.org", 443), ("files.pythonhosted.org", 443), ("1.1.1.1", 443)):
    try:
        socket.create_connection((h, p), timeout=4).close(); print("connect ok", h)
    except Exception as e:
        print("connect FAILED", h, type(e).__name__, e)
PY
} >> $L 2>&1
if PYTHONPATH=/tmp/cv python -c "import cv2" 2>/dev/null; then
  echo "== cv2 importable: running the fixture recipe" >> $L
  PYTHONPATH=/tmp/cv python tests/golden/make_opencv_fixtures.py >> $L 2>&1
  cp tests/golden/opencv_frames.npz tests/golden/opencv_frames.json gpurun_out/ 2>>$L
  PYTHONPATH=/tmp/cv python -m pytest tests/test_opencv_fixtures.py -q >> $L 2>&1
else
  echo "== no cv2 by any of the three probes: parity at the cv:: boundary stays unpinned" >> $L
fi
tail -40 $L
