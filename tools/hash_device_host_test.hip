// Host-side execution of the device hash routines (acvm_amd/csrc/hash_device.hpp is __host__ __device__): reads lines "func hex-message"
// (func: sha256 | blake2s | keccak256; "-" = the empty message) on stdin and prints the hex digest of each; tests/test_hash_device_on_host.py
// compares with hashlib and with the oracle's Keccak. No GPU is needed: hipcc builds the host side and nothing is launched.
#include "../acvm_amd/csrc/hash_device.hpp"
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
using namespace acvm;

struct HostMsg {  // the accessor the kernels' LdsMsg / MsgBuf implement
    const std::vector<uint8_t> *bytes;
    uint32_t word_le(uint32_t wi, uint32_t len) const {
        uint32_t v = 0;
        for (uint32_t k = 0; k < 4; k++) {
            const uint32_t at = 4 * wi + k;
            if (at < len) v |= (uint32_t)(*bytes)[at] << (8 * k);
        }
        return v;
    }
};

int main() {
    char func[32];
    static char hex[1 << 16];
    while (scanf("%31s %65535s", func, hex) == 2) {
        std::vector<uint8_t> msg;
        if (strcmp(hex, "-") != 0)
            for (size_t i = 0; hex[i] && hex[i + 1]; i += 2) {
                unsigned b = 0;
                sscanf(hex + i, "%2x", &b);
                msg.push_back((uint8_t)b);
            }
        const HostMsg m{&msg};
        Digest d;
        if (!strcmp(func, "sha256")) d = sha256_body(m, (uint32_t)msg.size());
        else if (!strcmp(func, "blake2s")) d = blake2s_body(m, (uint32_t)msg.size());
        else d = keccak256_body(m, (uint32_t)msg.size());
        for (uint32_t i = 0; i < 32; i++) printf("%02x", d.byte(i));
        printf("\n");
    }
    return 0;
}
