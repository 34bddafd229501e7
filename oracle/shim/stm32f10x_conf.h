/* TEST INFRASTRUCTURE ONLY -- intentionally empty (reference Src/BSP/ADC.H:4 includes it). */
