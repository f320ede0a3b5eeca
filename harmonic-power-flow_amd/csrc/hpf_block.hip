// placeholder: replaced by the block-tree solver
#include "hpf_internal.hpp"
namespace hpf {
int tree_build(hpf_handle*, const hpf_desc*) { return HPF_E_TOPOLOGY; }
void tree_free(hpf_handle*) {}
int tree_alloc_scenarios(hpf_handle*) { return HPF_OK; }
int tree_newton_step(hpf_handle*, bool) { return HPF_E_STATE; }
}
