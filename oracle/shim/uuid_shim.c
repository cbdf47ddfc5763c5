/* oracle/shim/uuid_shim.c -- TEST INFRASTRUCTURE ONLY (see uuid/uuid.h). */
#include "uuid/uuid.h"
static unsigned char g_fixed[16] = {0xC1,0x9E,0xF0,0x12,0x00,0x01,0x40,0x00,0x80,0x00,0xA3,0x55,0x0C,0xDA,0x40,0x00};
void uuid_generate(uuid_t out) { for (int i = 0; i < 16; i++) out[i] = g_fixed[i]; }
/* test hook: choose the GUID the reference will stamp into the next clip */
void oracle_set_guid(const unsigned char *g) { for (int i = 0; i < 16; i++) g_fixed[i] = g[i]; }
