/* TEST INFRASTRUCTURE ONLY -- typedef shim so the reference's Speech_Recog
 * sources compile on a host compiler.  Supplies the fixed-width names the
 * firmware gets from its vendor header (reference Src/StdPeriph_Driver/
 * stm32f10x.h:421-439) and an opaque USART_TypeDef so USART.H parses. */
#ifndef SR_ORACLE_SHIM_STM32F10X_H
#define SR_ORACLE_SHIM_STM32F10X_H
#include <stdint.h>
typedef uint8_t  u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef int8_t   s8;
typedef int16_t  s16;
typedef int32_t  s32;
typedef struct USART_TypeDef USART_TypeDef;
#endif
