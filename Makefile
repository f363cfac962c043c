# Build of the MI355X-native gmove path. `make` builds everything the tests/bench need:
#   poregen_amd/libpgmove.so      HIP kernels + C ABI (include/pgmove.h), gfx950 only
#   poregen_amd/_pg_hosttest.so   host-only build of the shared host/device arithmetic (CPU tests)
#   bin/poregen                   the drop-in `poregen gmove` CLI (host C++ over the C ABI)
#   oracle/                       the CPU oracle (test infrastructure)
HIPCC ?= /opt/rocm/bin/hipcc
CXX ?= g++
ARCH ?= gfx950
HIPFLAGS = --offload-arch=$(ARCH) -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-result
CSRC = poregen_amd/csrc

all: poregen_amd/libpgmove.so poregen_amd/_pg_hosttest.so bin/poregen oracle_build

# libpgmove.so deliberately does NOT carry a DT_NEEDED on libamdhip64: a process must hold exactly one HIP
# runtime, and under Python that has to be the copy PyTorch bundles (poregen_amd/_abi.py preloads it
# RTLD_GLOBAL); the CLI links /opt/rocm's libamdhip64 itself.
build/%.o: $(CSRC)/%.hip $(CSRC)/pg_internal.h $(CSRC)/pg_dev.h $(CSRC)/pg_select.h $(CSRC)/pg_model.h include/pgmove.h
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<

poregen_amd/libpgmove.so: build/pg_kernels.o build/pg_place.o build/pg_api.o build/pg_model.o build/pg_job.o build/pg_text.o
	$(CXX) -shared -o $@ $^ -Wl,--allow-shlib-undefined -ldl -lpthread

poregen_amd/_pg_hosttest.so: $(CSRC)/pg_hosttest.cpp $(CSRC)/pg_hostmem.h $(CSRC)/pg_select.h $(CSRC)/pg_model.h $(CSRC)/host/io.cpp $(CSRC)/host/dump.cpp $(CSRC)/host/pg_host.h
	$(CXX) -O2 -std=c++17 -fPIC -shared -ffp-contract=off -I$(CSRC) -o $@ $(CSRC)/pg_hosttest.cpp $(CSRC)/host/io.cpp $(CSRC)/host/dump.cpp -lz -lpthread

HOST = $(CSRC)/host
bin/poregen: $(CSRC)/pg_model.h $(HOST)/main.cpp $(HOST)/gmove_cli.cpp $(HOST)/reform_cli.cpp $(HOST)/io.cpp $(HOST)/dump.cpp $(HOST)/pg_host.h include/pgmove.h poregen_amd/libpgmove.so
	@mkdir -p bin
	$(CXX) -O2 -g -std=c++17 -Wall -o $@ $(HOST)/main.cpp $(HOST)/gmove_cli.cpp $(HOST)/reform_cli.cpp $(HOST)/io.cpp $(HOST)/dump.cpp \
	    -Lporegen_amd -lpgmove -L/opt/rocm/lib -lamdhip64 -lz -lpthread -Wl,-rpath,'$$ORIGIN/../poregen_amd' -Wl,-rpath,/opt/rocm/lib

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

clean:
	rm -f poregen_amd/libpgmove.so poregen_amd/_pg_hosttest.so
	$(MAKE) -C oracle clean

.PHONY: all clean oracle_build fallback_probe variant
