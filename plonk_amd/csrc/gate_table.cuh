// Gate evaluation table from the flattened trace: wire columns gathered by node id and one-hot
// selector columns by operation kind.
//
// Reference: build_gate_evaluation_table, src/execution_trace.rs:248-302.  As shipped it handles
// only the single-input tables (:268-296, "similar processing ... omitted", :298-299) and indexes
// vectors that were created with capacity but no length (:238-241, :286-289), so it cannot run;
// this is the completed form of the pattern it shows: for the row of operation `kind` at global row
// j (offset-table numbering, src/offset_table.rs:22-48)
//   input1[j] = get_val(Input(0)); input2[j] = get_val(Input(1)) if the operation has a second
//   input, else default_val (:287); input3[j] likewise for a third input (:288);
//   output[j] = get_val(Output) (:289); selectors[kind][j] = one (:292-294), zero in the other
//   selector columns; get_val(pos) = wire_map[pos], or default_val when pos is absent (:260-265).
// The wire map is flattened as wire_map[(j, w)] = node_values[node_ids[w][j]]: a wire carries its
// node's evaluated value (NodeMeta, src/ast.rs:6-9); a position whose node id is GT_NO_WIRE or not
// below the number of node values is absent from the map.
#pragma once
#include "field.cuh"

namespace plk {

constexpr uint64_t GT_NO_WIRE = ~0ull;
constexpr int GT_THREADS = 256;

struct GateTableCols {
  const uint64_t* node[4];   // node ids per wire, n each
  uint64_t* wire[4];         // out: input1, input2, input3, output
  uint64_t* sel[9];          // out: selector per ASTKind index, nullptr = kind not in the trace
};

// row_ops[j] = kind (low 4 bits, ASTKind index 0..8) | inputs (high 4 bits, 1..3)
__global__ __launch_bounds__(GT_THREADS) void gate_table_fill_kernel(GateTableCols cols, const uint8_t* row_ops, uint64_t n,
                                                                    const uint64_t* node_values, uint64_t m, Fe default_val, Fe one) {
  const uint64_t j = (uint64_t)blockIdx.x * GT_THREADS + threadIdx.x;
  if (j >= n) return;
  const uint32_t op = row_ops[j];
  const uint32_t kind = op & 15u, inputs = op >> 4;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const uint64_t id = cols.node[w][j];
    const bool wired = (w == 3 || (uint32_t)w < inputs) && id < m;   // GT_NO_WIRE is never below m
    fe_store(cols.wire[w], j, wired ? fe_load(node_values, id) : default_val);
  }
  const Fe zero = fe_zero();
#pragma unroll
  for (int k = 0; k < 9; ++k)
    if (cols.sel[k]) fe_store(cols.sel[k], j, kind == (uint32_t)k ? one : zero);
}

}  // namespace plk
