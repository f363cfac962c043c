# Build of the MI355X-native gmove path. `make` builds everything the tests/bench need:
#   poregen_amd/libpgmove.so      HIP kernels + C ABI (include/pgmove.h), gfx950 only
#   poregen_amd/_pg_hosttest.so   host-only build of the shared host/device arithmetic (CPU tests)
#   bin/poregen                   the drop-in `poregen gmove` CLI (host C++ over the C ABI)
#   oracle/                       the CPU oracle (test infrastructure)
HIPCC ?= /opt/rocm/bin/hipcc
CXX ?= g++
CC ?= gcc
ARCH ?= gfx950
HIPFLAGS = --offload-arch=$(ARCH) -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-result
CSRC = poregen_amd/csrc

all: poregen_amd/libpgmove.so poregen_amd/_pg_hosttest.so bin/poregen oracle_build

# libpgmove.so deliberately does NOT carry a DT_NEEDED on libamdhip64: a process must hold exactly one HIP
# runtime, and under Python that has to be the copy PyTorch bundles (poregen_amd/_abi.py preloads it
# RTLD_GLOBAL); the CLI links /opt/rocm's libamdhip64 itself.
build/%.o: $(CSRC)/%.hip $(CSRC)/pg_job_rule.h $(CSRC)/pg_internal.h $(CSRC)/pg_dev.h $(CSRC)/pg_select.h $(CSRC)/pg_model.h include/pgmove.h
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<

poregen_amd/libpgmove.so: build/pg_kernels.o build/pg_place.o build/pg_api.o build/pg_model.o build/pg_job.o build/pg_text.o
	$(CXX) -shared -o $@ $^ -Wl,--allow-shlib-undefined -ldl -lpthread

poregen_amd/_pg_hosttest.so: $(CSRC)/pg_hosttest.cpp $(CSRC)/pg_job_rule.h $(CSRC)/pg_hostmem.h $(CSRC)/pg_select.h $(CSRC)/pg_model.h $(CSRC)/host/io.cpp $(CSRC)/host/dump.cpp $(CSRC)/host/pg_host.h
	$(CXX) -O2 -std=c++17 -fPIC -shared -ffp-contract=off -I$(CSRC) -o $@ $(CSRC)/pg_hosttest.cpp $(CSRC)/host/io.cpp $(CSRC)/host/dump.cpp -lz -lpthread -ldl

HOST = $(CSRC)/host
bin/poregen: $(CSRC)/pg_model.h $(HOST)/main.cpp $(HOST)/gmove_cli.cpp $(HOST)/reform_cli.cpp $(HOST)/io.cpp $(HOST)/dump.cpp $(HOST)/pg_host.h include/pgmove.h poregen_amd/libpgmove.so
	@mkdir -p bin
	$(CXX) -O2 -g -std=c++17 -Wall -o $@ $(HOST)/main.cpp $(HOST)/gmove_cli.cpp $(HOST)/reform_cli.cpp $(HOST)/io.cpp $(HOST)/dump.cpp \
	    -Lporegen_amd -lpgmove -L/opt/rocm/lib -lamdhip64 -lz -lpthread -ldl -Wl,-rpath,'$$ORIGIN/../poregen_amd' -Wl,-rpath,/opt/rocm/lib

# measurement build: counts the reads whose selection leaves the fast path (tools/count_fallbacks.py)
fallback_probe:
	@mkdir -p build/fb
	for f in pg_kernels pg_place pg_api pg_model pg_job pg_text; do $(HIPCC) $(HIPFLAGS) -DPG_COUNT_FALLBACKS -c -o build/fb/$$f.o $(CSRC)/$$f.hip || exit 1; done
	$(CXX) -shared -o build/fb/libpgmove_fb.so build/fb/pg_kernels.o build/fb/pg_place.o build/fb/pg_api.o build/fb/pg_model.o build/fb/pg_job.o build/fb/pg_text.o -Wl,--allow-shlib-undefined

# A/B builds: `make variant NAME=x EXTRA="-DPG_..."` -> build/x/libpgmove.so (bench.py --lib, tools/ab_lib.sh)
variant:
	@mkdir -p build/$(NAME)
	for f in pg_kernels pg_place pg_api pg_model pg_job pg_text; do $(HIPCC) $(HIPFLAGS) $(EXTRA) -c -o build/$(NAME)/$$f.o $(CSRC)/$$f.hip || exit 1; done
	$(CXX) -shared -o build/$(NAME)/libpgmove.so build/$(NAME)/pg_kernels.o build/$(NAME)/pg_place.o build/$(NAME)/pg_api.o build/$(NAME)/pg_model.o build/$(NAME)/pg_job.o build/$(NAME)/pg_text.o -Wl,--allow-shlib-undefined -ldl -lpthread

oracle_build:
	$(MAKE) -C oracle

# Sanitizer builds of everything that runs on the HOST (the reference has `make asan=1`, /root/reference/Makefile:28-31, and a valgrind
# mode, test/test.sh:31-37): the shared host/device arithmetic + the file readers (build/asan/_pg_hosttest.so), the CPU oracle
# (build/asan/libgmove_oracle.so, gmove_oracle, model_oracle) and `poregen reform` alone (build/asan/poregen_reform: the reform subtool
# needs no device code). CPU box only -- there is no GPU AddressSanitizer on this pool. Run the CPU suite under them with
#     make asan && make asan_test
# (= LD_PRELOAD of libasan + PG_HOSTTEST_SO / PG_ORACLE_DIR / PG_REFORM_BIN pointing at build/asan, python's own leaks not reported).
SAN = -fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined -g -O1
asan:
	@mkdir -p build/asan
	$(CXX) $(SAN) -std=c++17 -fPIC -shared -ffp-contract=off -I$(CSRC) -o build/asan/_pg_hosttest.so $(CSRC)/pg_hosttest.cpp $(HOST)/io.cpp $(HOST)/dump.cpp -lz -lpthread -ldl
	$(CC) $(SAN) -std=gnu99 -fPIC -ffp-contract=off -shared -o build/asan/libgmove_oracle.so oracle/gmove_oracle.c -lm
	$(CC) $(SAN) -std=gnu99 -ffp-contract=off -o build/asan/gmove_oracle oracle/gmove_oracle_cli.c oracle/gmove_oracle.c -lm
	$(CC) $(SAN) -std=gnu99 -ffp-contract=off -o build/asan/model_oracle oracle/model_oracle.c -lm
	$(CXX) $(SAN) -std=c++17 -DPG_REFORM_ONLY -o build/asan/poregen_reform $(HOST)/main.cpp $(HOST)/reform_cli.cpp $(HOST)/io.cpp $(HOST)/dump.cpp -lz -lpthread -ldl
asan_test: asan
	ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 LD_PRELOAD=$$($(CC) -print-file-name=libasan.so):$$($(CC) -print-file-name=libubsan.so) \
	    PG_HOSTTEST_SO=$(CURDIR)/build/asan/_pg_hosttest.so PG_ORACLE_DIR=$(CURDIR)/build/asan PG_REFORM_BIN=$(CURDIR)/build/asan/poregen_reform POREGEN_CLEAN_EXIT=1 \
	    python3 -m pytest tests -x -q -m "not gpu" -p no:cacheprovider

clean:
	rm -f poregen_amd/libpgmove.so poregen_amd/_pg_hosttest.so
	$(MAKE) -C oracle clean

.PHONY: all clean oracle_build fallback_probe variant asan asan_test
