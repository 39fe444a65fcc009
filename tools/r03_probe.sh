mkdir -p gpurun_out/r03
HIPVOC_WINO=1 timeout -k 10 300 python3 tools/margin_stages.py hostile 0 0 16 > gpurun_out/r03/margin_stages_hostile_wino1.txt 2>&1 && \
HIPVOC_WINO=0 timeout -k 10 300 python3 tools/margin_stages.py hostile 0 0 16 > gpurun_out/r03/margin_stages_hostile_wino0.txt 2>&1 && \
timeout -k 10 300 python3 tools/margin_stages.py hostile 0 2 16 > gpurun_out/r03/margin_stages_hostile_x3.txt 2>&1 && \
timeout -k 10 300 python3 tools/margin_stages.py gauss 0 0 16 > gpurun_out/r03/margin_stages_gauss_wino1.txt 2>&1 && \
tools/gpu_tests.sh --deselect tests/test_gpu_margin.py::test_margin_over_seeds_and_a_hostile_family --deselect tests/test_gpu_margin.py::test_no_family_loses_more_than_10x_of_the_margin
