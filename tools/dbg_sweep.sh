for d in 0 1 2 3 4; do echo "DEBUG=$d"; SPUTNIK_HIP_SPMM_DEBUG=$d timeout -k 10 100 python tools/kbench.py --ops spmm --densities 0.25,0.1,0.05 2>&1 | grep '"op"' | cut -c60-140; done
