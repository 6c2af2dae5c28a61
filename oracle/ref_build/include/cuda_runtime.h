// TEST INFRASTRUCTURE (oracle/ref_build): empty stand-in for the CUDA header of this name; see ../cuda_on_cpu.h
#pragma once
