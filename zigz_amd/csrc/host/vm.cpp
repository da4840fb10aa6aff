// RV64IM interpreter + trace recording: the sequential front end of Prover.prove (src/vm/state.zig,
// src/vm/memory.zig, src/isa/rv64i.zig, src/isa/instruction_table.zig).  Host only by nature.
// Steps are recorded as compact 48-byte records (zigz_trace_step): the device rebuilds the 43 witness words of a step.
#include <cstring>
#include <memory>
#include <unordered_map>

#include "zigz_host.hpp"

namespace zigz {

enum : uint8_t {
    OP_LOAD = 0x03, OP_MISC_MEM = 0x0f, OP_OP_IMM = 0x13, OP_AUIPC = 0x17, OP_OP_IMM_32 = 0x1b, OP_STORE = 0x23,
    OP_OP = 0x33, OP_LUI = 0x37, OP_OP_32 = 0x3b, OP_BRANCH = 0x63, OP_JALR = 0x67, OP_JAL = 0x6f, OP_SYSTEM = 0x73,
    OP_LOAD_FP = 0x07, OP_STORE_FP = 0x27
};

static inline int64_t sext(uint32_t v, int bits) {
    const uint32_t m = 1u << (bits - 1);
    return (int64_t)(int32_t)((v ^ m) - m);
}

bool Instruction::decode(uint32_t w, Instruction &o) {  // rv64i.zig:124-233
    const uint8_t op = w & 0x7f;
    if (op == 0) return false;  // error.InvalidInstruction (:126-128)
    o.opcode = op;
    o.rd = (w >> 7) & 31;
    o.funct3 = (w >> 12) & 7;
    o.rs1 = (w >> 15) & 31;
    o.rs2 = (w >> 20) & 31;
    o.funct7 = (w >> 25) & 0x7f;
    switch (op) {  // instructionFormat, :61-73
    case OP_OP_IMM: case OP_OP_IMM_32: case OP_JALR: case OP_LOAD: case OP_LOAD_FP: case OP_MISC_MEM: case OP_SYSTEM:
        o.imm = sext(w >> 20, 12); break;
    case OP_STORE: case OP_STORE_FP:
        o.imm = sext(((w >> 25) << 5) | ((w >> 7) & 31), 12); break;
    case OP_BRANCH:
        o.imm = sext((((w >> 31) & 1) << 12) | (((w >> 7) & 1) << 11) | (((w >> 25) & 0x3f) << 5) | (((w >> 8) & 0xf) << 1), 13);
        break;
    case OP_LUI: case OP_AUIPC:
        o.imm = (int64_t)(int32_t)(w & 0xfffff000u); break;
    case OP_JAL:
        o.imm = sext((((w >> 31) & 1) << 20) | (((w >> 12) & 0xff) << 12) | (((w >> 20) & 1) << 11) | (((w >> 21) & 0x3ff) << 1), 21);
        break;
    default:
        o.imm = 0;  // R-type and unknown opcodes
    }
    return true;
}

bool hasLookupTable(const Instruction &in) {  // instruction_table.zig:243-274
    return in.opcode == OP_OP || in.opcode == OP_OP_IMM || in.opcode == OP_LOAD || in.opcode == OP_STORE ||
           in.opcode == OP_BRANCH;
}

// Sparse byte-addressable memory (memory.zig): unmapped bytes read 0.  Paged instead of per-byte hashed.
struct VMState::Mem {
    static constexpr uint64_t PAGE = 4096;
    std::unordered_map<uint64_t, std::unique_ptr<uint8_t[]>> pages;
    uint64_t last_tag = ~0ull;
    uint8_t *last = nullptr;
    uint64_t fetch_tag = ~0ull;  // page of the last instruction fetch (see fetch())
    const uint8_t *fetch_page = nullptr;
    uint8_t *page(uint64_t addr, bool create) {
        const uint64_t tag = addr / PAGE;
        if (tag == last_tag) return last;
        auto it = pages.find(tag);
        if (it == pages.end()) {
            if (!create) return nullptr;
            auto p = std::unique_ptr<uint8_t[]>(new uint8_t[PAGE]());
            it = pages.emplace(tag, std::move(p)).first;
            fetch_tag = ~0ull;  // a fetch from this page may have cached "unmapped"
        }
        last_tag = tag;
        last = it->second.get();
        return last;
    }
    uint8_t lb(uint64_t a) {
        uint8_t *p = page(a, false);
        return p ? p[a % PAGE] : 0;
    }
    void sb(uint64_t a, uint8_t v) { page(a, true)[a % PAGE] = v; }
    uint64_t load(uint64_t a, int n) {
        if ((a % PAGE) + (uint64_t)n <= PAGE) {  // inside one page (the common case): one copy, little-endian host
            const uint8_t *p = page(a, false);
            uint64_t v = 0;
            if (p) memcpy(&v, p + a % PAGE, (size_t)n);
            return v;
        }
        uint64_t v = 0;
        for (int b = 0; b < n; b++) v |= (uint64_t)lb(a + (uint64_t)b) << (8 * b);
        return v;
    }
    void store(uint64_t a, uint64_t v, int n) {
        if ((a % PAGE) + (uint64_t)n <= PAGE) {
            memcpy(page(a, true) + a % PAGE, &v, (size_t)n);
            return;
        }
        for (int b = 0; b < n; b++) sb(a + (uint64_t)b, (uint8_t)(v >> (8 * b)));
    }
    // instruction fetch: code and data alternate in a loop, so the page of the last fetch is remembered separately from
    // the page of the last data access
    uint32_t fetch(uint64_t pc) {
        if ((pc % PAGE) + 4 <= PAGE) {
            const uint64_t tag = pc / PAGE;
            if (tag != fetch_tag) {
                auto it = pages.find(tag);
                fetch_page = it == pages.end() ? nullptr : it->second.get();
                fetch_tag = tag;
            }
            uint32_t w = 0;
            if (fetch_page) memcpy(&w, fetch_page + pc % PAGE, 4);
            return w;
        }
        return (uint32_t)load(pc, 4);
    }
};

VMState::VMState(const std::vector<uint8_t> &program, uint64_t start_pc, const std::vector<uint64_t> *input)
    : pc(start_pc), mem_(new Mem()) {  // state.zig:72-93
    for (size_t i = 0; i < program.size(); i++) mem_->sb(start_pc + i, program[i]);
    if (input) input_tape_ = *input;
}
VMState::VMState(const std::vector<Segment> &segments, uint64_t entry_pc, const std::vector<uint64_t> *input)
    : pc(entry_pc), mem_(new Mem()) {  // state.zig:97-119
    for (auto &s : segments)
        for (size_t i = 0; i < s.data.size(); i++) mem_->sb(s.vaddr + i, s.data[i]);
    if (input) input_tape_ = *input;
}
VMState::~VMState() { delete mem_; }

uint64_t VMState::execute(const Instruction &in, uint64_t *mem_row) {  // state.zig:188-597
    const uint64_t a = readReg(in.rs1), b = readReg(in.rs2), imm = (uint64_t)in.imm;
    auto fail = [](int code, const char *name) -> uint64_t { throw Error(code, std::string("error.") + name); };
    switch (in.opcode) {
    case OP_OP: {
        uint64_t r;
        if (in.funct7 == 1) {  // RV64M, :226-286
            const int64_t sa = (int64_t)a, sb = (int64_t)b;
            switch (in.funct3) {
            case 0: r = a * b; break;
            case 1: r = (uint64_t)(int64_t)(((__int128)sa * (__int128)sb) >> 64); break;
            case 2: r = (uint64_t)(int64_t)(((__int128)sa * (__int128)(unsigned __int128)b) >> 64); break;
            case 3: r = (uint64_t)(((unsigned __int128)a * b) >> 64); break;
            case 4: r = sb == 0 ? ~0ull : (sa == INT64_MIN && sb == -1) ? a : (uint64_t)(sa / sb); break;
            case 5: r = b == 0 ? ~0ull : a / b; break;
            case 6: r = sb == 0 ? a : (sa == INT64_MIN && sb == -1) ? 0 : (uint64_t)(sa % sb); break;
            default: r = b == 0 ? a : a % b; break;
            }
        } else {
            const unsigned sh = (unsigned)(b & 63);
            switch (in.funct3) {
            case 0: r = in.funct7 == 0x20 ? a - b : a + b; break;
            case 1: r = a << sh; break;
            case 2: r = (int64_t)a < (int64_t)b; break;
            case 3: r = a < b; break;
            case 4: r = a ^ b; break;
            case 5: r = in.funct7 == 0x20 ? (uint64_t)((int64_t)a >> sh) : a >> sh; break;
            case 6: r = a | b; break;
            default: r = a & b; break;
            }
        }
        writeReg(in.rd, r);
        return pc + 4;
    }
    case OP_OP_32: {  // :319-397
        const uint32_t x = (uint32_t)a, y = (uint32_t)b;
        uint32_t r;
        if (in.funct7 == 1) {
            const int32_t sx = (int32_t)x, sy = (int32_t)y;
            switch (in.funct3) {
            case 0: r = x * y; break;
            case 4: r = sy == 0 ? 0xffffffffu : (sx == INT32_MIN && sy == -1) ? x : (uint32_t)(sx / sy); break;
            case 5: r = y == 0 ? 0xffffffffu : x / y; break;
            case 6: r = sy == 0 ? x : (sx == INT32_MIN && sy == -1) ? 0 : (uint32_t)(sx % sy); break;
            case 7: r = y == 0 ? x : x % y; break;
            default: return fail(ERR_INVALID_OP32, "InvalidOP32M");
            }
        } else {
            const unsigned sh = y & 31;
            switch (in.funct3) {
            case 0: r = in.funct7 == 0x20 ? x - y : x + y; break;
            case 1: r = x << sh; break;
            case 5: r = in.funct7 == 0x20 ? (uint32_t)((int32_t)x >> sh) : x >> sh; break;
            default: return fail(ERR_INVALID_OP32, "InvalidOP32");
            }
        }
        writeReg(in.rd, (uint64_t)(int64_t)(int32_t)r);
        return pc + 4;
    }
    case OP_OP_IMM: {  // :399-425
        const unsigned sh = (unsigned)(imm & 63);
        uint64_t r;
        switch (in.funct3) {
        case 0: r = a + imm; break;
        case 1: r = a << sh; break;
        case 2: r = (int64_t)a < in.imm; break;
        case 3: r = a < imm; break;
        case 4: r = a ^ imm; break;
        case 5: r = in.funct7 == 0x20 ? (uint64_t)((int64_t)a >> sh) : a >> sh; break;
        case 6: r = a | imm; break;
        default: r = a & imm; break;
        }
        writeReg(in.rd, r);
        return pc + 4;
    }
    case OP_OP_IMM_32: {  // :427-450
        const uint32_t x = (uint32_t)a;
        const unsigned sh = (unsigned)(imm & 31);
        uint32_t r;
        switch (in.funct3) {
        case 0: r = x + (uint32_t)imm; break;
        case 1: r = x << sh; break;
        case 5: r = in.funct7 == 0x20 ? (uint32_t)((int32_t)x >> sh) : x >> sh; break;
        default: return fail(ERR_INVALID_OP32, "InvalidOPIMM32");
        }
        writeReg(in.rd, (uint64_t)(int64_t)(int32_t)r);
        return pc + 4;
    }
    case OP_LOAD: {  // :452-482
        const uint64_t addr = a + imm;
        uint64_t r;
        switch (in.funct3) {
        case 0: r = (uint64_t)(int64_t)(int8_t)mem_->load(addr, 1); break;
        case 1: r = (uint64_t)(int64_t)(int16_t)mem_->load(addr, 2); break;
        case 2: r = (uint64_t)(int64_t)(int32_t)mem_->load(addr, 4); break;
        case 3: r = mem_->load(addr, 8); break;
        case 4: r = mem_->load(addr, 1); break;
        case 5: r = mem_->load(addr, 2); break;
        case 6: r = mem_->load(addr, 4); break;
        default: return fail(ERR_INVALID_LOAD_FUNCT3, "InvalidLoadFunct3");
        }
        mem_row[0] = addr; mem_row[1] = r; mem_row[2] = 1;  // access_type == .Load => is_read = 1
        writeReg(in.rd, r);
        return pc + 4;
    }
    case OP_STORE: {  // :484-507
        const uint64_t addr = a + imm;
        if (in.funct3 > 3) return fail(ERR_INVALID_STORE_FUNCT3, "InvalidStoreFunct3");
        mem_->store(addr, b, 1 << in.funct3);
        mem_row[0] = addr; mem_row[1] = b; mem_row[2] = 0;  // value = full rs2
        return pc + 4;
    }
    case OP_BRANCH: {  // :509-528
        bool t;
        switch (in.funct3) {
        case 0: t = a == b; break;
        case 1: t = a != b; break;
        case 4: t = (int64_t)a < (int64_t)b; break;
        case 5: t = (int64_t)a >= (int64_t)b; break;
        case 6: t = a < b; break;
        case 7: t = a >= b; break;
        default: return fail(ERR_INVALID_BRANCH_FUNCT3, "InvalidBranchFunct3");
        }
        return t ? pc + imm : pc + 4;
    }
    case OP_JAL: writeReg(in.rd, pc + 4); return pc + imm;
    case OP_JALR: { const uint64_t base = a; writeReg(in.rd, pc + 4); return (base + imm) & ~1ull; }
    case OP_LUI: writeReg(in.rd, imm); return pc + 4;
    case OP_AUIPC: writeReg(in.rd, pc + imm); return pc + 4;
    case OP_SYSTEM:  // :564-597
        if (in.funct3 == 0 && in.imm == 0) {
            const uint64_t sc = readReg(17);
            if (sc == 1) output_tape.push_back(readReg(10));  // ECALL_COMMIT
            else if (sc == 2) writeReg(10, input_pos_ < input_tape_.size() ? input_tape_[input_pos_++] : 0);  // ECALL_READ
            return pc + 4;
        }
        if (in.funct3 == 0 && in.imm == 1) { halted = true; return pc; }  // EBREAK
        return fail(ERR_UNIMPLEMENTED_SYSTEM, "UnimplementedSYSTEM");
    case OP_MISC_MEM: return pc + 4;  // FENCE, :202-205
    default: return fail(ERR_UNIMPLEMENTED_INSTRUCTION, "UnimplementedInstruction");
    }
}

void ExecutionTrace::reserveSteps(size_t n) {
    steps.reserve(n);
    is_lookup.reserve(n);
}

zigz_trace_step *ExecutionTrace::appendStep() {
    const size_t i = is_lookup.size();
    if (i + 1 > steps.size()) {
        // grow the logical size in blocks of 4096 steps so the per-step cost is one bounds check; records beyond
        // stepCount() are scratch until recorded (consumers use stepCount(), never steps.size())
        size_t want = i + 4096;
        if (want > steps.capacity()) steps.reserve(steps.capacity() * 2 > want ? steps.capacity() * 2 : want);
        steps.resize(want);
    }
    return steps.data() + i;
}

std::vector<uint64_t> ExecutionTrace::expandRows() const {
    const size_t ns = stepCount();
    std::vector<uint64_t> rows(ns * ROW_WORDS);
    uint64_t regs[32];
    memcpy(regs, initial_regs, sizeof(regs));
    regs[0] = 0;
    for (size_t i = 0; i < ns; i++) {
        const zigz_trace_step &st = steps[i];
        if (st.wr_reg && st.wr_reg < 32) regs[st.wr_reg] = st.rd_value;  // like the device: 0 and values >= 32 mean no write
        uint64_t *row = rows.data() + i * ROW_WORDS;
        row[0] = st.pc;
        memcpy(row + 1, regs, sizeof(regs));  // regs_after
        row[33] = st.opcode; row[34] = st.rd; row[35] = st.rs1; row[36] = st.rs2; row[37] = st.funct3; row[38] = st.funct7;
        row[39] = (uint64_t)st.imm;
        row[40] = st.mem_addr; row[41] = st.mem_value; row[42] = st.mem_is_read;
    }
    return rows;
}

void VMState::step() {  // state.zig:128-167
    if (halted) throw Error(ERR_VM_HALTED, "error.VMHalted");
    Instruction in;
    if (!Instruction::decode(mem_->fetch(pc), in)) {
        halted = true;
        invalid_instruction = true;  // error.InvalidInstruction: no step recorded
        return;
    }
    if (trace.stepCount() == 0) {  // register file before the first recorded step (initial_regs were written by now)
        memcpy(trace.initial_regs, regs_, sizeof(regs_));
        trace.initial_regs[0] = 0;
    }
    uint64_t mem_row[3] = {0, 0, 0};
    const uint64_t pc_before = pc;
    wr_reg_ = 0;  // every instruction writes at most one register (execute() below); x0 never changes
    wr_val_ = 0;
    const uint64_t next_pc = execute(in, mem_row);
    zigz_trace_step *st = trace.appendStep();
    st->pc = pc_before;
    st->rd_value = wr_val_;
    st->mem_addr = mem_row[0];
    st->mem_value = mem_row[1];
    st->imm = in.imm;
    st->opcode = in.opcode; st->rd = in.rd; st->rs1 = in.rs1; st->rs2 = in.rs2; st->funct3 = in.funct3; st->funct7 = in.funct7;
    st->wr_reg = wr_reg_;
    st->mem_is_read = (uint8_t)mem_row[2];
    trace.is_lookup.push_back(hasLookupTable(in) ? 1 : 0);
    pc = next_pc;
    step_count++;
}

void VMState::run(size_t max_steps) {  // state.zig:172-184
    size_t steps = 0;
    while (!halted && steps < max_steps) {
        step();
        if (invalid_instruction) return;
        steps++;
    }
    if (steps >= max_steps && !halted) throw Error(ERR_MAX_STEPS_EXCEEDED, "error.MaxStepsExceeded");
}

}  // namespace zigz
