// TEST INFRASTRUCTURE: x86-64 SysV cooperative stack switch used by tests/emu/hip_emu.h.
// emu_switch(void** save_sp, void* new_sp): push callee-saved registers, store rsp, load the other
// stack, pop its callee-saved registers, return into it.
__asm__(
    ".text\n"
    ".globl emu_switch\n"
    ".type emu_switch,@function\n"
    "emu_switch:\n"
    "    pushq %rbp\n"
    "    pushq %rbx\n"
    "    pushq %r12\n"
    "    pushq %r13\n"
    "    pushq %r14\n"
    "    pushq %r15\n"
    "    movq %rsp, (%rdi)\n"
    "    movq %rsi, %rsp\n"
    "    popq %r15\n"
    "    popq %r14\n"
    "    popq %r13\n"
    "    popq %r12\n"
    "    popq %rbx\n"
    "    popq %rbp\n"
    "    ret\n"
    ".size emu_switch,.-emu_switch\n");
