RT=$(python -c "from haphic_amd import build; print(build.asan_runtime())")
echo "RT=$RT"; ls -la haphic_amd/libhaphic_hip_asan.so
export LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:verify_asan_link_order=0 HAPHIC_HIP_SO=haphic_amd/libhaphic_hip_asan.so
python -c "print('py ok')" ; echo "rc=$?"
python -c "import torch; print('torch', torch.cuda.is_available())"; echo "rc=$?"
python -c "
import numpy as np
from haphic_amd import _lib
print('dev', _lib.device_count())
m=_lib.DeviceCSR.from_arrays(np.array([0,1,2],np.int32), np.array([0,1],np.int32), np.array([1,1],np.float32))
print(m.shape3)
"; echo "rc=$?"
python -m pytest tests/test_bam.py -q -x -p no:cacheprovider -m gpu 2>&1 | tail -5; echo "rc=$?"
