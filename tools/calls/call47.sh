#!/bin/bash
source tools/gpu_call.sh
bash tools/vb.sh "- variants/v_nopost.so variants/v_o2.so variants/v_nocluster.so" "65536"
