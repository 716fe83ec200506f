/* oracle/dropin_log_test.c - TEST INFRASTRUCTURE ONLY. A caller of the reference's PHY library that registers a log handler
 * (srslte_phy_log_register_handler, lib/src/phy/utils/phy_logger.c:37-52) and then provokes one diagnostic from a function that
 * libsrslte_phy_hip.so serves (srslte_ofdm_rx_init with an invalid bandwidth, ofdm.c:237-240): the message must arrive at the handler,
 * not on stderr. Built by ref_hip.mk against the reference's upper library + libsrslte_phy_hip.so; exit 0 = handler got the message. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "srslte/phy/dft/ofdm.h"
#include "srslte/phy/utils/phy_logger.h"

static int  got = 0;
static char last[256];
static void handler(phy_logger_level_t level, void* ctx, char* str)
{
  (void)ctx;
  if (level == LOG_LEVEL_ERROR_S) {
    got++;
    strncpy(last, str, sizeof(last) - 1);
  }
}

int main(void)
{
  srslte_ofdm_t q;
  cf_t*         a = calloc(30720, sizeof(cf_t));
  cf_t*         b = calloc(30720, sizeof(cf_t));
  srslte_phy_log_register_handler(NULL, handler);
  if (srslte_ofdm_rx_init(&q, SRSLTE_CP_NORM, a, b, 111) == SRSLTE_SUCCESS) {
    printf("srslte_ofdm_rx_init accepted 111 PRB\n");
    return 1;
  }
  printf("handler calls: %d, last: %s\n", got, last);
  return got >= 1 && strstr(last, "nof_prb") ? 0 : 2;
}
