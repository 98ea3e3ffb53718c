# Python-free build of libdxo_hip.so (the same flags dolfinx_external_operator_amd/_build.py uses) and of the C demo.
#   make            -> dolfinx_external_operator_amd/libdxo_hip.so
#   make demo       -> examples/c_abi_demo
HIPCC   ?= hipcc
ARCH    ?= gfx950
CSRC    := dolfinx_external_operator_amd/csrc
OBJDIR  := dolfinx_external_operator_amd/build
LIB     := dolfinx_external_operator_amd/libdxo_hip.so
SRCS    := $(wildcard $(CSRC)/*.hip)
OBJS    := $(patsubst $(CSRC)/%.hip,$(OBJDIR)/%.o,$(SRCS))
FLAGS   := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Iinclude -I$(CSRC)

# icnn.hip: no SLP re-packing of the neuron-pair fp32 arithmetic into v_pk_*_f32 (see the comment at icnn_f2 there)
$(OBJDIR)/icnn.o: FLAGS += -fno-slp-vectorize

all: $(LIB)

$(OBJDIR)/%.o: $(CSRC)/%.hip $(wildcard $(CSRC)/*.h) include/dxo.h
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(FLAGS) -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS) -ldl -lpthread

demo: $(LIB) examples/c_abi_demo.c
	gcc -O2 -Wall -Iinclude examples/c_abi_demo.c -o examples/c_abi_demo -Ldolfinx_external_operator_amd -ldxo_hip \
	    -Wl,-rpath,'$$ORIGIN/../dolfinx_external_operator_amd' -lm

clean:
	rm -rf $(OBJDIR) $(LIB) examples/c_abi_demo

.PHONY: all demo clean
