# Top-level convenience targets. The product is built by `python -m acvm_amd.build` (hipcc, gfx950); the oracle by oracle/Makefile.
#   make asan   the host code that parses untrusted bytes (circuit reader, WitnessMap reader, planner) with AddressSanitizer and
#               UndefinedBehaviorSanitizer, CPU only: tools/asan/fuzz_driver (tests/test_fuzz_reader.py feeds it mutated circuits)
CXX ?= g++
ASAN_FLAGS = -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -Wall -Wno-sign-compare
ASAN_SRCS = tools/asan/fuzz_driver.cpp acvm_amd/csrc/circuit.cpp acvm_amd/csrc/plan.cpp acvm_amd/csrc/tuning.cpp acvm_amd/csrc/schedule.cpp acvm_amd/csrc/schedule_check.cpp
asan: tools/asan/fuzz_driver
tools/asan/fuzz_driver: $(ASAN_SRCS) $(wildcard acvm_amd/csrc/*.hpp)
	$(CXX) $(ASAN_FLAGS) -o $@ $(ASAN_SRCS) -lz
.PHONY: asan
